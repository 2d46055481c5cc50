#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 300 python -m pytest tests/test_gpu_sync_free.py -q --tb=short -k graph 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30
timeout -s KILL 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
