#!/bin/bash
# K1 across workloads for two settings of its occupancy cap (LG_K1_DYN_LDS bytes of unused dynamic LDS; 0 = no cap)
run() { # label, env value, lib, bench args...
  local label=$1 dyn=$2 lib=$3; shift 3
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
  if [ "$dyn" = "def" ]; then unset LG_K1_DYN_LDS; else export LG_K1_DYN_LDS=$dyn; fi
  timeout -s KILL 300 python bench.py "$@" --no-cpu-baseline --no-literal 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python - "$label" "$dyn" "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json")); k = d.get("kernels_ms", {})
print(sys.argv[1], "dyn", sys.argv[2], sys.argv[3], "value", d["value"], "k1", k.get("preprocess"), "k9", k.get("preprocess_bwd"), flush=True)
PY
}
for rep in 1 2; do
run c3 0 - --mode fwdbwd --steps 50
run c3 def - --mode fwdbwd --steps 50
run c3 def variants/lib_k1_hoist.so --mode fwdbwd --steps 50
run c3 0 variants/lib_k1_hoist.so --mode fwdbwd --steps 50
done
run c2 0 - --mode fwd --n-gaussians 1000000 --steps 100
run c2 def - --mode fwd --n-gaussians 1000000 --steps 100
run c5 0 - --mode fwdbwd --n-gaussians 6000000 --width 1600 --height 1060 --sh-degree 2 --steps 30
run c5 def - --mode fwdbwd --n-gaussians 6000000 --width 1600 --height 1060 --sh-degree 2 --steps 30
run fwd6 0 - --mode fwd --n-gaussians 6000000 --width 1600 --height 1060 --steps 30
run fwd6 def - --mode fwd --n-gaussians 6000000 --width 1600 --height 1060 --steps 30
run heavy 0 - --mode fwdbwd --scene heavy --steps 30
run heavy def - --mode fwdbwd --scene heavy --steps 30
run count 0 - --mode count --steps 60
run count def - --mode count --steps 60
