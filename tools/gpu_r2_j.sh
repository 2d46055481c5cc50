#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show() { tail -1 $1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$2', d['value'], 'ms/step', d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('views_per_s'), d.get('heavier_scenes'), (d.get('roofline') or {}).get('traffic'), (d.get('roofline') or {}).get('valu_frac'), (d.get('roofline') or {}).get('profile'))
except Exception as e: print('RAW', l[-2500:])
"; }
for r in 1 2 3; do
  for m in off validated; do
    timeout -s KILL 300 python bench.py --steps 200 --warmup 20 --sync-free $m --no-cpu-baseline --no-literal --no-roofline > gpurun_out/r2j_$m.log 2>&1; show gpurun_out/r2j_$m.log sync_$m
  done
done
timeout -s KILL 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2j_full.log 2>&1; show gpurun_out/r2j_full.log full
