#!/bin/bash
# ordered kernel list of ONE fwd+bwd step (who issues the memsets / copies between our kernels)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_list -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-literal $LG_LISTING_ARGS > $R/gpurun_out/trace_list.log 2>&1
cd $R
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/trace_list/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if "lg_preprocess<" in r["Kernel_Name"]]
lo, hi = idx[-2], idx[-1]
prev_end = None
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    print('%7.1f us (+%5.1f gap)  %s' % ((e - s) / 1e3, gap, r['Kernel_Name'][:90]))
PY
rm -rf gpurun_out/trace_list
