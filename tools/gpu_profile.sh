#!/bin/bash
# Reproducible roofline evidence of ONE bench mode from the current build:  gpu_profile.sh <mode> [extra bench args]
#   pass 1  rocprofv3 --kernel-trace --stats            -> per-kernel launch time
#   pass 2/3 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes: TCC slots)  -> HBM traffic
#   pass 3-7 SQ counter sets                             -> instruction counts / busy cycles
# summarised by tools/profile_summary.py into gpurun_out/r04_profile_<mode>.json (copy to profiles/).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fwdbwd}; shift
B="python $R/bench.py --mode $MODE --no-cpu-baseline --no-roofline --no-literal $@"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$MODE -o s -- $B --steps 30 --warmup 5 > $R/gpurun_out/prof_$MODE.log 2>&1; echo "stats rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${MODE}_$i -o p -- $B --steps 3 --warmup 1 > $R/gpurun_out/pmc_${MODE}_$i.log 2>&1; echo "pmc $i ($set) rc=$?"
done
cd $R
S=$(find gpurun_out/prof_$MODE -name '*kernel_stats.csv' | head -1)
C() { find gpurun_out/pmc_${MODE}_$1 -name '*counter_collection.csv' | head -1; }
python tools/profile_summary.py --mode $MODE --stats $S --fetch $(C 1) --write $(C 2) --sq $(C 3) $(C 4) $(C 5) $(C 6) $(C 7) \
  --command "rocprofv3 [--kernel-trace --stats | --pmc <set>] -- python bench.py --mode $MODE --no-cpu-baseline --no-roofline --no-literal $* (tools/gpu_profile.sh)" \
  > gpurun_out/${PROFILE_TAG:-r06}_profile_$MODE.json
cp $S gpurun_out/${PROFILE_TAG:-r06}_${MODE}_kernel_stats.csv
head -c 1200 gpurun_out/${PROFILE_TAG:-r06}_profile_$MODE.json; echo
# the summary and the stats CSV are what travels back (gpurun merges at most 64 MiB); the raw rocprofv3 output stays on the box
rm -rf gpurun_out/prof_$MODE gpurun_out/pmc_${MODE}_[0-9]*
