#!/bin/bash
# A/B on ONE box (box-to-box variance is ~5 %): usage  gpu_ab.sh "<bench args>" libA.so libB.so ...   (two rounds, interleaved)
mkdir -p gpurun_out
ARGS="$1"; shift
for round in 1 2; do
  for lib in "$@"; do
    LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib timeout 300 python bench.py $ARGS --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$lib', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'))
except Exception as e: print('RAW', l[-1500:])
" | tee -a gpurun_out/ab.log
  done
done
