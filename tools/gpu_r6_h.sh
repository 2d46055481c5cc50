#!/bin/bash
# round 6, call H: the tile-level LDS merge applied to the INTEGER-weight significance pass (-DLG_MERGE_INT): one count atomic per (tile, entry)
# instead of one per (wave, entry) -- parity with the variant library, then A/B on one box
mkdir -p gpurun_out; export TMPDIR=/tmp
V=$PWD/lightgaussian_amd/variants/lib_mint.so
LIGHTGAUSSIAN_HIP_LIB=$V timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_tiles.py tests/test_gpu_sync_free.py tests/test_gpu_round5.py -m gpu -q --tb=short -x -k "count or long or sharded or fuzz or overflow" 2>&1 | grep -E "passed|failed" | tail -2
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline --n-gaussians 3000000 --mode count --steps 100 "$@" 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); print('${LIGHTGAUSSIAN_HIP_LIB##*/}', '$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', d.get('kernels_ms'), d['significance_pass']['mask_sha256'][:12])" | cut -c1-500; }
for rep in 1 2 3; do
for lib in - mint; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/lightgaussian_amd/variants/lib_$lib.so; fi
  run
done
done
for lib in - mint; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/lightgaussian_amd/variants/lib_$lib.so; fi
  run --scene heavy
done
