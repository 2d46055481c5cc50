#!/bin/bash
export TMPDIR=/tmp
run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); c=d['config']; print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', 'V', c['visible_gaussians'], 'R', c['tile_instances'], d.get('kernels_ms'))
except Exception as e: print('FAILED $*', l[-1500:])"; }
run --n-gaussians 1000000 --scale 0.02 --mode fwdbwd --steps 30
run --n-gaussians 1000000 --scale 0.06 --mode fwdbwd --steps 20
run --n-gaussians 200000 --scale 0.3 --mode fwdbwd --steps 10
run --n-gaussians 1000000 --scale 0.02 --mode count --steps 30
