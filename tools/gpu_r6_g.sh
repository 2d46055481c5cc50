#!/bin/bash
# round 6, call G: the tile-level LDS merge of the per-hit count variant (one plain store per instance instead of an atomic per (wave, entry))
# against the previous build (lib_prev) and with ring sizes 2 / 8 (default 4); parity tests first
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_weight_policies.py tests/test_gpu_long_tiles.py tests/test_gpu_sync_free.py -m gpu -q --tb=short -x 2>&1 | tail -3 > gpurun_out/g_pytest.log; cat gpurun_out/g_pytest.log
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline --n-gaussians 3000000 --mode count --steps 100 "$@" 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); print('${LIGHTGAUSSIAN_HIP_LIB##*/}', '$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', d.get('kernels_ms'))" | cut -c1-500; }
for rep in 1 2; do
for lib in - prev tm2 tm8; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/lightgaussian_amd/variants/lib_$lib.so; fi
  run --weight-policy alpha_t
done
done
unset LIGHTGAUSSIAN_HIP_LIB
run --weight-policy alpha
run --weight-policy alpha_t --scene heavy
run
