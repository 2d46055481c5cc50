#!/bin/bash
# round 2, first GPU call: new kernels first (sort / VQ / compaction / sync-free), then the whole -m gpu suite, smoke(), benches,
# and a rocprofv3 kernel-stats pass of the bench command.  Every step under its own timeout.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T="--timeout 180"
for f in test_gpu_sort test_gpu_vq test_gpu_compact test_gpu_sync_free test_gpu_prune_epilogue test_gpu_dropin_replay; do
  timeout 600 python -m pytest tests/$f.py -m gpu -q --tb=short $T 2>&1 | tail -40 > gpurun_out/r2a_$f.log
  echo "== $f: $(grep -E 'passed|failed|error' gpurun_out/r2a_$f.log | tail -1)"
done
timeout 1200 python -m pytest tests -m gpu -q --tb=short $T 2>&1 | tail -80 > gpurun_out/r2a_pytest_gpu.log
echo "== ALL: $(grep -E 'passed|failed|error' gpurun_out/r2a_pytest_gpu.log | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r2a_smoke.log
for mode in fwdbwd fwd count; do
  timeout 400 python bench.py --steps 100 --warmup 10 --mode $mode > gpurun_out/r2a_bench_$mode.log 2>&1
  tail -1 gpurun_out/r2a_bench_$mode.log | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$mode', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), d.get('camera_batch_3'), d.get('significance_pass'))
except Exception as e: print('RAW', l[-3000:])
"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2a_prof -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-literal > $R/gpurun_out/r2a_prof.log 2>&1
cd $R
f=$(find gpurun_out/r2a_prof -name '*kernel_stats.csv' | head -1); echo $f; head -30 $f
find gpurun_out/r2a_prof -name '*kernel_trace.csv' -size +20M -delete
find gpurun_out/r2a_prof -name '*.db' -size +20M -delete
