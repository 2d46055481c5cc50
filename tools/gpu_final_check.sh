#!/bin/bash
# what the driver runs at round end: gpu tests, smoke, default bench, torchrun launch form
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E 'passed|failed' | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; tail -4 gpurun_out/bench_default.log | cut -c1-2500
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
timeout 300 python examples/significance_prune.py 2>&1 | tail -1 | cut -c1-300
timeout 300 python examples/finetune_step.py 2>&1 | tail -1 | cut -c1-300
