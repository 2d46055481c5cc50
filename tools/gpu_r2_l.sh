#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
LG_FUZZ_N2=500 timeout -s KILL 500 python tools/gpu_fuzz.py 0 2>&1 | tail -8
timeout -s KILL 600 python -m pytest tests -m gpu -q 2>&1 | tail -3
