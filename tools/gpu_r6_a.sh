#!/bin/bash
# round 6, call A: the per-hit weight policies (Q24.40 sums) -- parity tests + the significance pass per policy
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); c=d['config']; sp=d.get('significance_pass') or {}; print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', d.get('kernels_ms'), {k: sp.get(k) for k in ('weight_policy','score_checksum','mask_sha256')})" | cut -c1-900; }
timeout -s KILL 900 python -m pytest tests/test_gpu_weight_policies.py tests/test_gpu_parity.py tests/test_gpu_prune_epilogue.py tests/test_gpu_sync_free.py tests/test_gpu_long_tiles.py -m gpu -q --tb=short -x 2>&1 | tail -4
run --n-gaussians 3000000 --mode count --steps 100
run --n-gaussians 3000000 --mode count --steps 100
run --n-gaussians 3000000 --mode count --steps 100 --weight-policy alpha_t
run --n-gaussians 3000000 --mode count --steps 100 --scene heavy
