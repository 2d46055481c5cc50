#!/bin/bash
# staged run C: full -m gpu suite, smoke, benches (default / validated sync-free / count / fwd), each step guarded
mkdir -p gpurun_out; export TMPDIR=/tmp
alive() { timeout -s KILL 60 python -c "import torch; print('gpu ok', float(torch.ones(4, device='cuda').sum()))" 2>&1 | tail -1; }
step() { local name=$1 t=$2; shift 2
  timeout -s KILL $t "$@" > gpurun_out/r2c_$name.log 2>&1; local rc=$?
  echo "== $name rc=$rc: $(grep -E 'passed|failed|error|Error' gpurun_out/r2c_$name.log | tail -2 | tr '\n' ' ' | cut -c1-300)"
  local a=$(alive); case "$a" in *"gpu ok"*) ;; *) echo "GPU NOT RESPONDING after $name -- stopping"; tail -20 gpurun_out/r2c_$name.log; exit 7;; esac
}
show() { tail -1 gpurun_out/r2c_$1.log | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$1', d['value'], 'ms/step', d['ms_per_step'], d.get('kernels_ms'), 'batch3', d.get('camera_batch_3'), 'steady', d.get('steady_state'), d.get('significance_pass'))
except Exception as e: print('RAW', l[-2500:])
"; }
step sort 200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_long_tiles.py tests/test_gpu_sync_free.py -q --tb=short
grep -E "^FAILED|^E  " gpurun_out/r2c_sort.log | head -20
step all 900 python -m pytest tests -m gpu -q --tb=short -x
grep -E "^FAILED|^E  " gpurun_out/r2c_all.log | head -20
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"; tail -2 gpurun_out/r2c_smoke.log
step bench_default 300 python bench.py --steps 100 --warmup 10; show bench_default
step bench_validated 300 python bench.py --steps 100 --warmup 10 --sync-free validated --no-cpu-baseline --no-literal; show bench_validated
step bench_count 300 python bench.py --steps 100 --warmup 10 --mode count --no-cpu-baseline; show bench_count
step bench_fwd 300 python bench.py --steps 100 --warmup 10 --mode fwd --no-cpu-baseline --no-literal; show bench_fwd
step tile_stats_heavy 200 python tools/tile_stats.py 3000000 heavy; tail -5 gpurun_out/r2c_tile_stats_heavy.log
step bench_heavy 300 python bench.py --steps 50 --warmup 5 --scene heavy --no-cpu-baseline --no-literal; show bench_heavy
