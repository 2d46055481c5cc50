#!/bin/bash
# PMC counters for our kernels (one pass per counter set; each wrapped in timeout: rocprofv3+torch can hang at exit)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc$i -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-literal > $R/gpurun_out/pmc$i.log 2>&1
  echo "set $i rc=$?"
  find $R/gpurun_out/pmc$i -name '*.csv' | head
done
cd $R
python - <<'PY'
import glob, csv, collections
for f in sorted(glob.glob('gpurun_out/pmc*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:28]
        if not k.startswith(('lg_', 'void lg_')): continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    print(f)
    for k, d in agg.items(): print('  ', k, {c: round(v) for c, v in d.items()})
PY
