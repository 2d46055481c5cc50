#!/bin/bash
# A/B of library variants on one box: tools/gpu_ab.sh <mode> <lib-a> <lib-b> ...   ("-" = the in-tree library)
mode=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
  timeout -s KILL 300 python bench.py --mode $mode --no-cpu-baseline --no-literal 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d.get("kernels_ms", {})
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "steady", d.get("steady_state", {}).get("views_per_s"),
      "bwd", k.get("blend_bwd"), "fwd", k.get("blend_fwd"), "k1", k.get("preprocess"), "k9", k.get("preprocess_bwd"), "sort", k.get("sort"), "tsort", k.get("tile_sort"), "dup", k.get("duplicate"), "batch3", d.get("camera_batch_3", {}).get("views_per_s_per_gpu"), flush=True)
PY
done
done
