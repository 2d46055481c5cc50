#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show() { tail -1 $1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$2', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), 'batch3', (d.get('camera_batch_3') or {}).get('views_per_s_per_gpu'), 'steady', (d.get('steady_state') or {}).get('views_per_s'))
except Exception as e: print('RAW', l[-2500:])
"; }
timeout -s KILL 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > gpurun_out/r2f_all.log; grep -E "passed|failed" gpurun_out/r2f_all.log | tail -1; grep -E "^FAILED|^E  " gpurun_out/r2f_all.log | head
timeout -s KILL 120 python -c "import torch; print('gpu ok', float(torch.ones(4, device='cuda').sum()))" | tail -1
for lib in lightgaussian_amd/liblightgaussian_hip.so lightgaussian_amd/variants/lib_d11.so lightgaussian_amd/liblightgaussian_hip.so lightgaussian_amd/variants/lib_d11.so; do
  LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib timeout -s KILL 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-literal > gpurun_out/r2f_b.log 2>&1; show gpurun_out/r2f_b.log $lib
done
for lib in lightgaussian_amd/liblightgaussian_hip.so lightgaussian_amd/variants/lib_d11.so; do
  LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib timeout -s KILL 300 python bench.py --steps 50 --warmup 5 --scene heavy --no-cpu-baseline --no-literal > gpurun_out/r2f_h.log 2>&1; show gpurun_out/r2f_h.log heavy:$lib
  LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -q -k "dropped_depth or fuzz or forward_count" 2>&1 | tail -2
done
