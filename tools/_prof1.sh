export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for extra in ""; do
rm -rf /tmp/p1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $R/bench.py --mode fwdbwd --no-cpu-baseline --no-roofline --no-literal --steps 30 --warmup 5 $extra > /tmp/p1.log 2>&1
S=$(find /tmp/p1 -name '*kernel_stats.csv' | head -1)
echo "== $extra"; python - "$S" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
done
