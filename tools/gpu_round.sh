#!/bin/bash
# One round's measurement run on a GPU box: the whole -m gpu suite, smoke, the bench modes / configs quoted in DESIGN.md 11, the
# rocprofv3 evidence (tools/gpu_profile.sh -> gpurun_out/<round>_profile_<mode>.json + kernel stats CSV; copy them to profiles/).
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); c=d['config']; print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', 'burst', (d.get('contract_region') or {}).get('views_per_s'), 'V', c['visible_gaussians'], 'R', c['tile_instances'], d.get('kernels_ms'), 'batch3', (d.get('camera_batch_3') or {}), (d.get('significance_pass') or {}), (d.get('literal_getter_pattern') or {}))" | cut -c1-1800; }
timeout -s KILL 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > gpurun_out/round_pytest.log; grep -E "passed|failed" gpurun_out/round_pytest.log | tail -1; grep -E "^FAILED|^E  " gpurun_out/round_pytest.log | cut -c1-300 | head
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout -s KILL 900 python bench.py ) > gpurun_out/round_bench_default.log 2>&1; tail -5 gpurun_out/round_bench_default.log | cut -c1-7000
run --n-gaussians 1000000 --mode fwd --steps 100 --no-literal
run --n-gaussians 3000000 --mode fwd --steps 100 --no-literal
run --n-gaussians 3000000 --mode count --steps 100
run --n-gaussians 3000000 --mode count --steps 100 --weight-policy alpha_t
run --n-gaussians 3000000 --mode count --steps 100 --weight-policy alpha
run --n-gaussians 3000000 --mode count --steps 100 --scene heavy
run --n-gaussians 3000000 --mode count --steps 100 --scene heavy --weight-policy alpha_t
run --n-gaussians 3000000 --mode count --steps 100 --scene heavy --count-streams 1
run --n-gaussians 3000000 --mode count --steps 100 --scene heavy --count-streams 1 --count-long-tiles parallel --segment-length 2048
run --n-gaussians 3000000 --mode count --steps 100 --scale 0.0045
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --no-fuse
run --n-gaussians 3000000 --mode fwdbwd --steps 100 --exact-exp --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 100 --sync-free off --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 100 --long-tiles serial --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --loss l1_dssim --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --loss l1_dssim_lazy --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --loss l1_dssim --loss-item --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --loss l1_dssim_lazy --loss-item --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 30 --loss l1_dssim_torch --no-literal
run --n-gaussians 6000000 --width 1600 --height 1060 --mode fwdbwd --steps 50 --sh-degree 2 --no-literal
run --n-gaussians 6000000 --width 1600 --height 1060 --mode fwd --steps 50 --sh-degree 3 --no-literal
run --n-gaussians 6000000 --width 1600 --height 1060 --mode distill --steps 30 --sh-degree 3
run --n-gaussians 3000000 --mode fwdbwd --steps 30 --scale 0.012 --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 50 --scene heavy --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 50 --scene heavy --long-tiles serial --no-literal
run --n-gaussians 3000000 --mode fwdbwd --steps 50 --scale 0.0045 --no-literal
run --n-gaussians 6000000 --width 3840 --height 2160 --mode fwdbwd --steps 20 --no-literal
timeout -s KILL 300 python tools/vq_bench.py 2>&1 | tail -4
timeout -s KILL 420 python tools/gpu_fuzz.py 150 2>&1 | tail -2 | cut -c1-300
timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
# the data-parallel step and the C4 leg of `bench.py --gpus N` through RCCL at world size 1 (the collectives are real, the wire is not)
timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-literal --force-collectives 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('forced collectives:', d['value'], 'views/s', d.get('data_parallel'), d.get('c4_significance_pass'))" | cut -c1-1500
for v in "--dense-allreduce" "--visible-allreduce" "--views-per-rank 4" "--dp-overlap"; do
timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-literal --no-roofline --no-c4-leg --force-collectives $v 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); dp = d.get('data_parallel') or {}; print('forced collectives $v:', d['value'], 'views/s', d['ms_per_step'], 'ms; exchange', dp.get('allreduce_ms'), 'ms, bytes on wire', dp.get('bytes_on_wire_per_step'))" | cut -c1-400
done
timeout -s KILL 300 python examples/significance_prune.py 2>&1 | tail -1 | cut -c1-300
timeout -s KILL 300 python examples/finetune_step.py 2>&1 | tail -1 | cut -c1-300
timeout -s KILL 300 python examples/finetune_step.py --fused-adam 2>&1 | tail -1 | cut -c1-300
timeout -s KILL 300 python examples/finetune_step.py --n-gaussians 3000000 --iters 40 2>&1 | tail -1 | cut -c1-300
timeout -s KILL 300 python examples/finetune_step.py --n-gaussians 3000000 --iters 40 --fused-adam 2>&1 | tail -1 | cut -c1-300
run --n-gaussians 3000000 --mode fwdbwd --steps 60 --spatial-order --no-literal
bash tools/gpu_profile.sh fwdbwd 2>&1 | tail -3 | cut -c1-200
bash tools/gpu_profile.sh count --count-streams 1 2>&1 | tail -3 | cut -c1-200
