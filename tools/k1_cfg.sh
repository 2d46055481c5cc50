#!/bin/bash
# K1 across workloads for library variants:  tools/k1_cfg.sh - variants/lib_k1_nocap.so ...   ("-" = the in-tree library)
#   e.g. the occupancy cap of round 4:  make -C lightgaussian_amd/csrc OUT=../../variants/lib_k1_nocap.so EXTRA="-DLG_K1_PAD_LDS=0"
run() { # label, lib, bench args...
  local label=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
  timeout -s KILL 300 python bench.py "$@" --no-cpu-baseline --no-literal 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python - "$label" "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json")); k = d.get("kernels_ms", {})
print(sys.argv[1], sys.argv[2], "value", d["value"], "k1", k.get("preprocess"), "k9", k.get("preprocess_bwd"), flush=True)
PY
}
mkdir -p gpurun_out
for lib in "$@"; do
run c3 $lib --mode fwdbwd --steps 50
run c3 $lib --mode fwdbwd --steps 50
run c2 $lib --mode fwd --n-gaussians 1000000 --steps 100
run c5 $lib --mode fwdbwd --n-gaussians 6000000 --width 1600 --height 1060 --sh-degree 2 --steps 30
run fwd6 $lib --mode fwd --n-gaussians 6000000 --width 1600 --height 1060 --steps 30
run heavy $lib --mode fwdbwd --scene heavy --steps 30
run count $lib --mode count --steps 60
done
