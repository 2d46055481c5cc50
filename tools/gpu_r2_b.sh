#!/bin/bash
# Staged, defensive GPU run: each step in its own process under a hard kill timeout; after every step a liveness probe of
# the GPU -- if it fails the script stops at once (a hung GPU must not be waited on until the lease dies).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
alive() { timeout -s KILL 60 python -c "import torch; print('gpu ok', float(torch.ones(4, device='cuda').sum()))" 2>&1 | tail -1; }
step() { # name, timeout, command...
  local name=$1 t=$2; shift 2
  timeout -s KILL $t "$@" > gpurun_out/r2b_$name.log 2>&1; local rc=$?
  echo "== $name rc=$rc: $(grep -E 'passed|failed|error|Error' gpurun_out/r2b_$name.log | tail -2 | tr '\n' ' ')"
  local a=$(alive); echo "   $a"
  case "$a" in *"gpu ok"*) ;; *) echo "GPU NOT RESPONDING after $name -- stopping"; tail -20 gpurun_out/r2b_$name.log; dmesg 2>/dev/null | tail -20; exit 7;; esac
}
alive
step sort_small 120 python -m pytest tests/test_gpu_sort.py -q --tb=short -k "1-0-8-0 or 63-3 or 8192-0 or 8193-22"
step sort_all 180 python -m pytest tests/test_gpu_sort.py -q --tb=short
tail -25 gpurun_out/r2b_sort_all.log
step parity 400 python -m pytest tests/test_gpu_parity.py -q --tb=short -x
tail -25 gpurun_out/r2b_parity.log
for f in test_gpu_vq test_gpu_compact test_gpu_prune_epilogue test_gpu_dropin_replay test_gpu_sync_free test_gpu_long_tiles; do
  step $f 300 python -m pytest tests/$f.py -q --tb=short
  grep -E "^FAILED|^ERROR|assert|Error" gpurun_out/r2b_$f.log | head -12
done
step bench 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline
tail -1 gpurun_out/r2b_bench.log | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('fwdbwd', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), d.get('camera_batch_3'), d.get('steady_state'))
except Exception as e: print('RAW', l[-3000:])
"
