#!/bin/bash
# round 6, call F: the heavier scenes once more on another box (box-to-box spread of the HBM-bound kernels)
for a in "--scale 0.012 --steps 30" "--scene heavy --steps 50" "--steps 100"; do
  timeout -s KILL 300 python bench.py --no-cpu-baseline --no-literal --n-gaussians 3000000 --mode fwdbwd $a 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d.get('kernels_ms'))"
done
