#!/bin/bash
# first GPU contact: parity tests, smoke, a short bench, rocprof stats
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log | tail -40
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_first.log 2>&1; tail -3 gpurun_out/bench_first.log
