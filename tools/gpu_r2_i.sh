#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show() { tail -1 $1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$2', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), 'steady', (d.get('steady_state') or {}).get('views_per_s'))
except Exception as e: print('RAW', l[-2500:])
"; }
bash tools/gpu_ab.sh "--steps 100 --warmup 10 --no-literal" lightgaussian_amd/liblightgaussian_hip.so lightgaussian_amd/variants/lib_dup512.so 2>&1 | cut -c1-400
timeout -s KILL 200 python tools/tile_stats.py 3000000 heavy 2>&1 | tail -4
timeout -s KILL 300 python bench.py --steps 50 --warmup 5 --scene heavy --no-cpu-baseline --no-literal > gpurun_out/r2i_h.log 2>&1; show gpurun_out/r2i_h.log heavy
timeout -s KILL 300 python bench.py --steps 50 --warmup 5 --scale 0.0055 --no-cpu-baseline --no-literal > gpurun_out/r2i_u.log 2>&1; show gpurun_out/r2i_u.log uniform_scale0.0055
bash tools/gpu_profile.sh fwdbwd 2>&1 | tail -12 | cut -c1-300
bash tools/gpu_profile.sh count 2>&1 | tail -8 | cut -c1-300
