#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "--- sort variants"
for lib in "" lightgaussian_amd/variants/lib_w1.so lightgaussian_amd/variants/lib_w4.so lightgaussian_amd/variants/lib_s512.so lightgaussian_amd/variants/lib_s512i16.so lightgaussian_amd/variants/lib_s1024i4.so lightgaussian_amd/variants/lib_s256i16.so; do
  if [ -n "$lib" ]; then export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; else unset LIGHTGAUSSIAN_HIP_LIB; fi
  timeout -s KILL 120 python tools/sort_bench.py 2>&1 | tail -1
  timeout -s KILL 120 python tools/sort_bench.py 11300000 29 61 2>&1 | tail -1
done
unset LIGHTGAUSSIAN_HIP_LIB
echo "--- new tests"
timeout -s KILL 300 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_loss.py tests/test_gpu_parity.py -q --tb=short 2>&1 | tail -8
echo "--- bench A/B (default build vs K7 without the T select)"
bash tools/gpu_ab.sh "--steps 100 --warmup 10 --no-literal --sync-free validated" lightgaussian_amd/liblightgaussian_hip.so lightgaussian_amd/variants/lib_notsel.so 2>&1 | cut -c1-400
echo "--- gradient equality of the two K7 variants"
timeout -s KILL 200 python - <<'PY'
import os, subprocess, sys, json
code = r'''
import math, torch, sys
sys.path.insert(0, ".")
from lightgaussian_amd import synthetic as syn
from lightgaussian_amd.gaussian_renderer import render
dev = torch.device("cuda:0")
g = syn.make_gaussians(200000, seed=3, log_scale_mean=math.log(0.02)).to(dev).requires_grad_(True)
cam = syn.orbit_camera(1, 7, 640, 360).to(dev)
gimg = torch.randn(3, 360, 640, generator=torch.Generator().manual_seed(1)).to(dev)
(render(cam, g, syn.PipelineParams(), torch.zeros(3, device=dev))["render"] * gimg).sum().backward()
torch.save({n: getattr(g, n).grad.cpu() for n in ("_xyz", "_features_rest", "_scaling", "_rotation", "_opacity")}, sys.argv[1])
'''
for name, lib in (("a", "lightgaussian_amd/liblightgaussian_hip.so"), ("b", "lightgaussian_amd/variants/lib_notsel.so")):
    subprocess.check_call([sys.executable, "-c", code, f"/tmp/g_{name}.pt"], env=dict(os.environ, LIGHTGAUSSIAN_HIP_LIB=os.path.abspath(lib)))
import torch
a, b = torch.load("/tmp/g_a.pt"), torch.load("/tmp/g_b.pt")
for k in a:
    print(k, "bit-identical" if torch.equal(a[k], b[k]) else f"max rel diff {float((a[k]-b[k]).abs().max() / a[k].abs().max()):.3e}")
PY
