#!/bin/bash
# HBM traffic of our kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), per the
# MI355X guide; raw values are KiB-units per dispatch; the gfx950 correction (FETCH_SIZE reports 1/2 of wide
# coalesced reads) is applied by tools/traffic_summary.py, not here.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fwdbwd}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o p -- python $R/bench.py --steps 3 --warmup 1 --mode $MODE --no-cpu-baseline --no-roofline --no-literal > $R/gpurun_out/pmc_$C.log 2>&1
  echo "$C rc=$?"
done
cd $R
python tools/traffic_summary.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv > gpurun_out/traffic_$MODE.json
cat gpurun_out/traffic_$MODE.json
rm -f gpurun_out/pmc_*/p_kernel_trace.csv
