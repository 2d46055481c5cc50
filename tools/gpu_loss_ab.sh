#!/bin/bash
# A/B of the fused L1 + SSIM kernels on one box: tools/gpu_loss_ab.sh <lib-a> <lib-b> ...   ("-" = the in-tree library)
# per library: the trainers' step (render fwd + L1/DSSIM + bwd) with the per-kernel times of the profile switch
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
  timeout -s KILL 300 python bench.py --mode fwdbwd --loss l1_dssim --steps 60 --no-cpu-baseline --no-literal 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d.get("kernels_ms", {})
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "steady", d.get("steady_state", {}).get("views_per_s"),
      "loss_fwd", k.get("loss_fwd"), "loss_bwd", k.get("loss_bwd"), "bwd", k.get("blend_bwd"), "fwd", k.get("blend_fwd"), flush=True)
PY
done
done
