#!/bin/bash
# Rehearsal of the driver's 8-GPU command at FULL size on a 1-GPU box: eight ranks share cuda:0, collectives over gloo (bench.py's test
# mode, LG_BENCH_SHARE_GPU=1).  Not a measurement -- it shows that the N = 8 line is produced, with which fields, and that the C4 leg's
# mask and the step's gradients agree across eight ranks at 3 M Gaussians / 1080p.
mkdir -p gpurun_out
export LG_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
W=${1:-8}
shift
TAG=${W}ranks_one_gpu$(echo "$*" | tr -c "a-zA-Z0-9\n" "_" | sed "s/_*$//; s/^_*/_/; s/^_$//")
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $W --backend gloo \
    --steps 4 --warmup 1 --no-cpu-baseline --no-literal "$@" > gpurun_out/r05_rehearsal_${TAG}.json 2> gpurun_out/r05_rehearsal_${TAG}.err
echo rc=$?
tail -c 1500 gpurun_out/r05_rehearsal_${TAG}.err
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r05_rehearsal_${TAG}.json") if l.startswith("{")][-1])
print({k: j.get(k) for k in ("metric","value","n_gpus","steps","ms_per_step","scaling","test_mode","gradients_identical_on_all_ranks")})
print(j.get("data_parallel"))
c=j.get("c4_significance_pass") or {}
print({k: c.get(k) for k in ("views","views_per_rank","mask_identical_on_all_ranks","mask_equals_1gpu","counts_equal_1gpu","scores_bit_identical_1gpu","views_per_s")})
P
