#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 300 python -X faulthandler -m pytest tests/test_gpu_sync_free.py -q --tb=short -k graph > gpurun_out/r2n_graph.log 2>&1
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|dist-packages/_pytest\|dist-packages/pluggy" gpurun_out/r2n_graph.log | head -60
