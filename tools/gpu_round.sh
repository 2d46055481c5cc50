#!/bin/bash
# GPU box: full -m gpu suite, smoke(), then short benches of the given modes (fwdbwd fwd count fused)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
for mode in "$@"; do
  EXTRA=""; if [ "$mode" = "nofuse" ]; then mode=fwdbwd; EXTRA="--no-fuse"; fi
  timeout 300 python bench.py --steps 50 --warmup 10 --mode $mode $EXTRA --no-cpu-baseline > gpurun_out/bench_$mode$EXTRA.log 2>&1; tail -1 gpurun_out/bench_$mode$EXTRA.log | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d['metric'], d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'))
except Exception as e: print('RAW', l[-2000:])
"
done
echo "PYTEST: $(grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -1)"; echo "SMOKE: $(tail -1 gpurun_out/smoke.log)"
