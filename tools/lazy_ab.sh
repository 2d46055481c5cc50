#!/bin/bash
python -m pytest tests/test_gpu_loss.py -x -q 2>&1 | tail -3
run() { timeout -s KILL 300 python bench.py --no-cpu-baseline --no-literal --no-roofline --mode fwdbwd --steps 60 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms')" | cut -c1-300; }
for rep in 1 2; do
run --loss l1_dssim
run --loss l1_dssim_lazy
run --loss l1_dssim --loss-item
run --loss l1_dssim_lazy --loss-item
done
