#!/bin/bash
# Round 5, first GPU call: the new tests, then the measurements that decide what stays.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r5_a.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r5a.log; : > $L
say() { echo "$@" | tee -a $L; }
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/r5a.err | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); c=d['config']; print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', 'burst', (d.get('contract_region') or {}).get('views_per_s'), 'V', c['visible_gaussians'], 'R', c['tile_instances'], d.get('kernels_ms'), (d.get('steady_state') or {}).get('distinct_cameras'), (d.get('significance_pass') or {}).get('views'))" | cut -c1-900 | tee -a $L; }
say "== new tests"
timeout -s KILL 900 python -m pytest tests/test_gpu_round5.py -q --tb=short -x 2>&1 | tail -25 | tee -a $L
say "== whole suite"
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -25 > gpurun_out/r5a_pytest.log; grep -E "passed|failed" gpurun_out/r5a_pytest.log | tail -1 | tee -a $L; grep -E "^FAILED|^E  " gpurun_out/r5a_pytest.log | cut -c1-300 | head -20 | tee -a $L
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $L
say "== K6 pair step: scalar lane masks (in-tree) vs the bool form (variant)"
for rep in 1 2; do
  for lib in - lightgaussian_amd/variants/lib_k6_bool.so; do
    if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
    say "lib $lib"; run --mode fwdbwd --steps 100 --no-literal; run --mode count --steps 100; run --mode count --steps 100 --scene heavy
  done
done
unset LIGHTGAUSSIAN_HIP_LIB
say "== K7 splat-parallel prototype vs the product kernel, three scenes"
for sc in "" "--scale 0.012" "--scene heavy"; do
  run --mode fwdbwd --steps 40 --no-literal $sc
  run --mode fwdbwd --steps 40 --no-literal $sc --bwd-splat-parallel
done
say "== data-parallel step through RCCL at world size 1 (collectives forced)"
dp() { timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-literal --no-roofline --force-collectives --no-c4-leg "${@:2}" 2>>gpurun_out/r5a.err | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${*:2}', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', json.dumps(d.get('data_parallel'))[:700])" | tee -a $L; }
dp 29521
dp 29522 --dense-allreduce
dp 29523 --visible-allreduce
dp 29524 --views-per-rank 4
dp 29525 --dp-overlap
say "== default bench line"
( time timeout -s KILL 900 python bench.py ) > gpurun_out/r5a_bench_default.log 2>&1; tail -4 gpurun_out/r5a_bench_default.log | cut -c1-3000 | tee -a $L
say "== rocprof of the prototype"
PROFILE_TAG=r05_proto_k7_splat_parallel bash tools/gpu_profile.sh fwdbwd --bwd-splat-parallel 2>&1 | tail -3 | cut -c1-300 | tee -a $L
