#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show() { tail -1 $1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$2', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), 'batch3', (d.get('camera_batch_3') or {}).get('views_per_s_per_gpu'), 'steady', (d.get('steady_state') or {}).get('views_per_s'))
except Exception as e: print('RAW', l[-2500:])
"; }
timeout -s KILL 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > gpurun_out/r2g_all.log; grep -E "passed|failed" gpurun_out/r2g_all.log | tail -1; grep -E "^FAILED|^E  " gpurun_out/r2g_all.log | head
timeout -s KILL 120 python -c "import torch; print('gpu ok', float(torch.ones(4, device='cuda').sum()))" | tail -1
timeout -s KILL 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-literal > gpurun_out/r2g_b.log 2>&1; show gpurun_out/r2g_b.log default
timeout -s KILL 300 python bench.py --steps 100 --warmup 10 --mode count --no-cpu-baseline > gpurun_out/r2g_c.log 2>&1; show gpurun_out/r2g_c.log count
