#!/bin/bash
# kernel trace of a short bench run (timeout-wrapped: rocprofv3 + torch may hang at exit after writing its output)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fwdbwd}
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$MODE -o t -- python $R/bench.py --steps 20 --warmup 5 --mode $MODE --no-cpu-baseline --no-roofline --no-literal > $R/gpurun_out/trace_$MODE.log 2>&1
echo "rc=$?"
cd $R
ls -la gpurun_out/trace_$MODE | head
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/trace_$MODE/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 10 steps worth: find the periodic structure by the preprocess kernel
idx = [i for i, r in enumerate(rows) if ('lg_preprocess<' in r['Kernel_Name'] or r['Kernel_Name'].startswith('lg_preprocess('))]
print('kernels', len(rows), 'preprocess launches', len(idx))
lo, hi = idx[-11], idx[-1]
seg = rows[lo:hi]
t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[hi]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print('10 steps: wall %.3f ms/step, kernel-busy %.3f ms/step, idle %.3f ms/step, %d kernels/step' % ((t1 - t0) / 1e7, busy / 1e7, (t1 - t0 - busy) / 1e7, len(seg) / 10))
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in seg:
    n = r['Kernel_Name'][:60]; agg[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e7; cnt[n] += 1
for n, v in sorted(agg.items(), key=lambda kv: -kv[1])[:28]: print('%8.4f ms/step x%-4.1f %s' % (v, cnt[n] / 10, n))
# biggest gaps
gaps = []
for a, b in zip(seg[:-1], seg[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    gaps.append((g / 1e3, a['Kernel_Name'][:40], b['Kernel_Name'][:40]))
gaps.sort(reverse=True)
print('largest gaps (us):')
for g in gaps[:12]: print('  %8.1f  %s -> %s' % g)
PY
# keep the small stats file only
rm -f gpurun_out/trace_$MODE/t_kernel_trace.csv
