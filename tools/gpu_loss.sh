#!/bin/bash
# GPU box: loss-kernel parity tests, then the fwd+bwd step with the three loss variants (bench.py --loss ...)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_loss.py -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_loss.log
for l in l1 l1_dssim l1_dssim_torch; do
  timeout 400 python bench.py --steps 40 --warmup 8 --loss $l --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_loss_$l.log | cut -c1-200
  python - <<PY
import json
j=json.loads(open("gpurun_out/bench_loss_$l.log").read())
print("$l", j["value"], "views/s", j["ms_per_step"], "ms", {k:v for k,v in j.get("kernels_ms",{}).items() if "loss" in k})
PY
done
