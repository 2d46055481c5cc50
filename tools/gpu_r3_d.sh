#!/bin/bash
# round 3, call D: reference-pinned kNN tests, runner tests, K6 A/B (in-tree = counted walk + single compare; sdot5 = round-2 K6)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_knn.py tests/test_gpu_dropin_runner.py tests/test_gpu_sort.py tests/test_gpu_sync_free.py -m gpu -q --tb=short 2>&1 | tail -30 > gpurun_out/r3d_pytest.log; grep -E "passed|failed|error" gpurun_out/r3d_pytest.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r3d_pytest.log | cut -c1-600 | head -30
bash tools/gpu_ab.sh fwdbwd lightgaussian_amd/variants/lib_sdot5.so - 
bash tools/gpu_ab.sh count lightgaussian_amd/variants/lib_sdot5.so -
