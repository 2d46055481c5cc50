#!/bin/bash
# parity tests + rocprofv3 kernel stats of the bench command + bench in the three modes
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
for mode in fwdbwd fwd count; do
  timeout 600 python bench.py --steps 50 --warmup 10 --mode $mode --no-cpu-baseline > gpurun_out/bench_$mode.log 2>&1; tail -1 gpurun_out/bench_$mode.log
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fwdbwd -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-literal > $GRAFT_REPO_ROOT/gpurun_out/prof_fwdbwd.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_fwdbwd | head -20
f=$(find gpurun_out/prof_fwdbwd -name '*kernel_stats.csv' | head -1); echo $f; head -25 $f
# keep only the small stats files
find gpurun_out/prof_fwdbwd -name '*kernel_trace.csv' -size +20M -delete
