#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for a in 0 1 2 3 4; do
  LG_ABLATE=$a timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); print('ABL=$a', d['value'], d['kernels_ms'].get('blend_bwd'))"
done
