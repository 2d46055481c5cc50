#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 500 python tools/gpu_fuzz.py 300 2>&1 | tail -8
LG_FUZZ_SEG=64 timeout -s KILL 400 python tools/gpu_fuzz.py 200 2>&1 | tail -6
LG_FUZZ_SYNC=off timeout -s KILL 300 python tools/gpu_fuzz.py 100 2>&1 | tail -4
