#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r5c.log; : > $L
say() { echo "$@" | tee -a $L; }
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline --no-roofline "$@" 2>>gpurun_out/r5c.err | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); c=d['config']; print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms')" | cut -c1-300 | tee -a $L; }
timeout -s KILL 900 python -m pytest tests/test_gpu_round5.py -q --tb=short -k "count" 2>&1 | tail -30 | tee -a $L
for st in 1 2 4; do
  run --mode count --steps 60 --scene heavy --count-streams $st --long-tiles serial
  run --mode count --steps 60 --scene heavy --count-streams $st --long-tiles parallel
done
run --mode count --steps 60 --count-streams 1 --long-tiles serial
run --mode count --steps 60 --count-streams 1 --long-tiles parallel
