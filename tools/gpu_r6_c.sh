#!/bin/bash
# round 6, call C: the new full-size parity cases (long lists) + the dp / count-long-tiles changes on the device
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 1700 python -m pytest tests/test_gpu_full_size.py -m gpu -q --tb=short -x -k "long_lists or alpha_t" -s 2>&1 | grep -v "^\[Gloo\]" | tail -25
timeout -s KILL 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_dp_runner.py tests/test_gpu_long_tiles.py -m gpu -q --tb=short -x 2>&1 | tail -5
