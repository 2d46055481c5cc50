#!/bin/bash
# BASELINE.json configs on one GPU (numbers for the results table)
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout 600 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); c=d['config']; print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', 'V', c['visible_gaussians'], 'R', c['tile_instances'], d.get('kernels_ms'))"; }
run --n-gaussians 1000000 --mode fwd --steps 100
run --n-gaussians 3000000 --mode fwd --steps 100
run --n-gaussians 3000000 --mode fwdbwd --steps 100
run --n-gaussians 3000000 --mode count --steps 100
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --no-fuse
run --n-gaussians 1000000 --mode fwdbwd --steps 100
run --n-gaussians 3000000 --mode fwdbwd --steps 100 --exact-exp --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --loss l1_dssim --no-literal
run --n-gaussians 6000000 --width 1600 --height 1060 --mode fwdbwd --steps 50 --sh-degree 2
run --n-gaussians 6000000 --width 1600 --height 1060 --mode fwd --steps 50 --sh-degree 3
run --n-gaussians 6000000 --width 1600 --height 1060 --mode distill --steps 30 --sh-degree 3
run --n-gaussians 6000000 --width 1600 --height 1060 --mode distill --steps 30 --sh-degree 3 --no-fuse
# heavier scenes (not BASELINE configs): 3x larger splats -> ~9x the tile instances
run --n-gaussians 3000000 --mode fwdbwd --steps 30 --scale 0.012 --no-literal
