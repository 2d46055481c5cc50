#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show() { tail -1 $1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$2', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), 'steady', (d.get('steady_state') or {}).get('views_per_s'))
except Exception as e: print('RAW', l[-2500:])
"; }
for lib in "" lightgaussian_amd/variants/lib_noforce.so lightgaussian_amd/variants/lib_w1.so lightgaussian_amd/variants/lib_s512.so; do
  if [ -n "$lib" ]; then export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; else unset LIGHTGAUSSIAN_HIP_LIB; fi
  timeout -s KILL 120 python tools/sort_bench.py 2>&1 | tail -1
done
unset LIGHTGAUSSIAN_HIP_LIB
timeout -s KILL 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > gpurun_out/r2h_all.log; grep -E "passed|failed" gpurun_out/r2h_all.log | tail -1; grep -E "^FAILED|^E  " gpurun_out/r2h_all.log | head
timeout -s KILL 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-literal > gpurun_out/r2h_b.log 2>&1; show gpurun_out/r2h_b.log default
timeout -s KILL 300 python bench.py --steps 100 --warmup 10 --mode fwd --no-cpu-baseline --no-literal > gpurun_out/r2h_f.log 2>&1; show gpurun_out/r2h_f.log fwd
timeout -s KILL 300 python bench.py --steps 50 --warmup 5 --scene heavy --no-cpu-baseline --no-literal > gpurun_out/r2h_h.log 2>&1; show gpurun_out/r2h_h.log heavy
