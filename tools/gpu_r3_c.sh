#!/bin/bash
# round 3, call C: the tests that failed in call B (after their fixes), then A/B of K7 variants on one box
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_long_tiles.py tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -40 > gpurun_out/r3c_pytest.log; grep -E "passed|failed|error" gpurun_out/r3c_pytest.log | tail -2; grep -E "^FAILED|^E  |C3 grad|C5 student" gpurun_out/r3c_pytest.log | cut -c1-900 | head -30
timeout -s KILL 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q --tb=line -s 2>&1 | grep -E "parity" | cut -c1-1500
bash tools/gpu_ab.sh fwdbwd lightgaussian_amd/variants/lib_base.so lightgaussian_amd/variants/lib_sdot5.so lightgaussian_amd/variants/lib_sdot6.so
