#!/bin/bash
# run D: full -m gpu suite (no -x), benches, and the reproducible profile pass of the fwdbwd mode (tools/gpu_profile.sh)
mkdir -p gpurun_out; export TMPDIR=/tmp
alive() { timeout -s KILL 60 python -c "import torch; print('gpu ok', float(torch.ones(4, device='cuda').sum()))" 2>&1 | tail -1; }
step() { local name=$1 t=$2; shift 2
  timeout -s KILL $t "$@" > gpurun_out/r2d_$name.log 2>&1; local rc=$?
  echo "== $name rc=$rc: $(grep -E 'passed|failed|error|Error' gpurun_out/r2d_$name.log | tail -2 | tr '\n' ' ' | cut -c1-300)"
  local a=$(alive); case "$a" in *"gpu ok"*) ;; *) echo "GPU NOT RESPONDING after $name -- stopping"; tail -20 gpurun_out/r2d_$name.log; exit 7;; esac
}
show() { tail -1 gpurun_out/r2d_$1.log | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$1', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), 'batch3', (d.get('camera_batch_3') or {}).get('views_per_s_per_gpu'), 'steady', (d.get('steady_state') or {}).get('views_per_s'))
except Exception as e: print('RAW', l[-2500:])
"; }
step all 900 python -m pytest tests -m gpu -q --tb=short
grep -E "^FAILED|^E  " gpurun_out/r2d_all.log | head -20
step bench_default 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline; show bench_default
step bench_validated 300 python bench.py --steps 100 --warmup 10 --sync-free validated --no-cpu-baseline --no-literal; show bench_validated
step profile 900 bash tools/gpu_profile.sh fwdbwd --sync-free validated
tail -40 gpurun_out/r2d_profile.log | cut -c1-400
python - <<'PY'
import json
j=json.load(open('gpurun_out/r02_profile_fwdbwd.json'))
for k,v in sorted(j['kernels'].items(), key=lambda kv:-kv[1].get('total_ns',0))[:16]:
    print(k[:60].ljust(60), 'avg_us', round(v.get('avg_ns',0)/1e3,1), 'calls', v.get('calls'), 'fetch', v.get('fetch_bytes_raw'), 'write', v.get('write_bytes_raw'), 'valu', v.get('SQ_INSTS_VALU'), 'busy', v.get('SQ_BUSY_CYCLES'), 'waves', v.get('SQ_WAVES'))
PY
