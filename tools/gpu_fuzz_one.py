#!/usr/bin/env python3
"""One trial of tools/gpu_fuzz.py looked at closely: python tools/gpu_fuzz_one.py TRIAL
Image of the training forward (hardware exp) and of the canonical arithmetic against the float32 oracle, pixel by pixel, with the
oracle's own contributor count and final transmittance at the worst pixels -- tells a stop-threshold flip (one pixel, a jump of up
to 1e-4 |colour|) from accumulated rounding (many pixels, small)."""
import sys

import numpy as np

import gpu_fuzz as F                                    # (imported: builds nothing, runs no trial)
from gpu_fuzz import gpu_common, oracle
from lightgaussian_amd import rasterizer

t = int(sys.argv[1])
kw, npk, meta, rs = F.make_trial(t)
ref = oracle.forward(count=True, **npk)
print("trial", t, meta)
for name, fast in (("hardware exp", True), ("canonical", False)):
    rasterizer.set_option("fast_exp", fast)
    img = gpu_common.hip_forward_backward(kw)["color"]
    d = np.abs(img - ref.color).max(axis=0)             # [H, W]
    order = np.argsort(d.ravel())[::-1][:3]
    print(f"{name}: max |image - oracle| {d.max():.3e}; pixels beyond 2e-6: {int((d > 2e-6).sum())} of {d.size}; beyond 1e-5: {int((d > 1e-5).sum())}")
    for p in order:
        y, x = divmod(int(p), meta["W"])
        print(f"    pixel ({x},{y}): diff {d[y, x]:.3e}  oracle final_T {ref.saved['final_T'][p]:.6e}  oracle n_contrib {int(ref.saved['n_contrib'][p])}")
rasterizer.set_option("fast_exp", True)
