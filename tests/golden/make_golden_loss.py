"""Generates tests/golden/reference_loss.npz by running the REFERENCE's own utils/loss_utils.py (plain torch, CPU)
in this container.  The fixture travels; /root/reference does not.

    python tests/golden/make_golden_loss.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference"
spec = importlib.util.spec_from_file_location("ref_loss_utils", os.path.join(REF, "utils", "loss_utils.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

LAMBDA = 0.2  # arguments/__init__.py lambda_dssim default
out = {"window": ref.gaussian(11, 1.5).numpy(), "window2d": ref.create_window(11, 3)[0, 0].numpy(), "lambda": np.float32(LAMBDA)}
gen = torch.Generator().manual_seed(20250103)
cases = {
    "rand_3x37x53": (3, 37, 53),      # ragged: not a multiple of any tile, wider than one 32-pixel tile
    "small_3x7x5": (3, 7, 5),         # smaller than the window: everything is padding
    "one_1x11x11": (1, 11, 11),
    "tile_edge_3x32x64": (3, 32, 64), # exact tile multiples
    "tall_2x70x9": (2, 70, 9),
}
for name, shp in cases.items():
    x = torch.rand(shp, generator=gen)
    y = (x + 0.1 * torch.randn(shp, generator=gen)).clamp(0, 1)
    if name.startswith("one"):
        y = x.clone()                 # identical images: ssim = 1, |x-y| = 0 everywhere (sign(0) = 0 gradient)
    x.requires_grad_(True)
    l1 = ref.l1_loss(x, y)
    ss = ref.ssim(x, y)
    loss = (1.0 - LAMBDA) * l1 + LAMBDA * (1.0 - ss)
    loss.backward()
    out[f"{name}.x"] = x.detach().numpy(); out[f"{name}.y"] = y.numpy()
    out[f"{name}.l1"] = l1.detach().numpy(); out[f"{name}.ssim"] = ss.detach().numpy()
    out[f"{name}.loss"] = loss.detach().numpy(); out[f"{name}.grad"] = x.grad.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_loss.npz"), **out)
print("wrote reference_loss.npz", {k: v.shape for k, v in out.items() if k.endswith(".x")})
