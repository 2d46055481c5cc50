#!/bin/bash
# round 6, call D: the two K7 experiments of the r5 verdict (item 4) with the kill criterion K7 <= 0.555 ms at C3:
#  (a) two contributing entries per reduction pass (-DLG_K7_PAIR_REDUCE, variant library) -- parity first, then A/B on one box
#  (b) the work items one level finer (segment length 384 / 256 against the default 512: every item of a long tile is split further)
mkdir -p gpurun_out; export TMPDIR=/tmp
V=$PWD/lightgaussian_amd/variants/lib_pair.so
LIGHTGAUSSIAN_HIP_LIB=$V timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_tiles.py -m gpu -q --tb=short -x -k "backward or long or segment" 2>&1 | tail -4
LIGHTGAUSSIAN_HIP_LIB=$V timeout -s KILL 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q --tb=short -x -k "c3_full_size" 2>&1 | tail -3
bash tools/gpu_ab.sh fwdbwd - lightgaussian_amd/variants/lib_pair.so
for S in 256 384 512 768; do
  timeout -s KILL 300 python bench.py --mode fwdbwd --no-cpu-baseline --no-literal --segment-length $S 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels_ms', {}); print('S=$S value', d['value'], 'ms', d['ms_per_step'], 'bwd', k.get('blend_bwd'), 'fwd', k.get('blend_fwd'))"
done
