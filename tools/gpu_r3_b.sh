#!/bin/bash
# round 3, call B: whole -m gpu suite (no -x), fuzz in the modes touched this round
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 > gpurun_out/r3b_pytest.log; grep -E "passed|failed|error" gpurun_out/r3b_pytest.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r3b_pytest.log | cut -c1-400 | head -40
timeout -s KILL 300 python tools/gpu_fuzz.py 120 2>&1 | tail -2
LG_NARROW_KEY=1 timeout -s KILL 300 python tools/gpu_fuzz.py 120 2>&1 | tail -2
LG_FUZZ_SEG=64 LG_FUZZ_LONG=parallel timeout -s KILL 300 python tools/gpu_fuzz.py 120 2>&1 | tail -2
LG_FUZZ_SEG=64 LG_FUZZ_LONG=auto timeout -s KILL 300 python tools/gpu_fuzz.py 120 2>&1 | tail -2
timeout -s KILL 300 python bench.py --no-cpu-baseline --no-literal --steps 100 2>&1 | tail -1 | cut -c1-400
