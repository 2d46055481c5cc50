#!/bin/bash
# parity tests + one short bench per requested mode (args: modes...)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
for mode in "$@"; do
  EXTRA=""; if [ "$mode" = "nofuse" ]; then mode=fwdbwd; EXTRA="--no-fuse"; fi
  timeout 300 python bench.py --steps 50 --warmup 10 --mode $mode $EXTRA --no-cpu-baseline > gpurun_out/bench_$mode$EXTRA.log 2>&1; tail -1 gpurun_out/bench_$mode$EXTRA.log | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d['metric'], d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), d.get('roofline'))
except Exception as e: print('RAW', l[-2000:])
"
done
