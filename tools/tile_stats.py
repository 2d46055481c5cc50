#!/usr/bin/env python3
"""Per-tile list-length and depth-of-traversal statistics of the bench workload (load balance of the per-tile kernels)."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import synthetic as syn
from lightgaussian_amd.gaussian_renderer import render

N, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 1920, 1080
dev = torch.device("cuda:0")
g_cpu = syn.make_gaussians(N)
if "heavy" in sys.argv:      # tile_stats.py <N> heavy : the heavy-tailed variant (bench.py --scene heavy)
    syn.make_heavy_tailed(g_cpu)
pc = g_cpu.to(dev).requires_grad_(True)
cam = syn.orbit_camera(0, 200, W, H).to(dev)
pkg = render(cam, pc, syn.PipelineParams(), torch.zeros(3, device=dev))
saved = pkg["render"].grad_fn.saved_tensors
binning, img = saved[-2], saved[-1]     # (..., radii, geom, binning, img) in both autograd Functions
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
ranges = binning[: T * 8].view(torch.int32).view(T, 2).cpu().numpy()
n = (ranges[:, 1] - ranges[:, 0]).reshape(gy, gx)
P = W * H
off = ((P * 4 + 255) // 256) * 256
ncontrib = img[off: off + P * 4].view(torch.int32).view(H, W).cpu().numpy()
pad = np.zeros((gy * 16, gx * 16), np.int32); pad[:H, :W] = ncontrib
wmax = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3))
print("tiles", T, "list length: mean %.0f  min %d  max %d  p99 %.0f  std %.0f" % (n.mean(), n.min(), n.max(), np.percentile(n, 99), n.std()))
print("deepest contributor per tile (what K7 walks): mean %.0f max %d; fraction of list walked %.2f" % (wmax.mean(), wmax.max(), wmax.sum() / max(n.sum(), 1)))
rows = n.mean(axis=1)
print("mean list length per tile row (top->bottom, every 8th):", np.round(rows[::8]).astype(int).tolist())
print("sum over the 8 contiguous bands:", [int(b.sum()) for b in np.array_split(n.reshape(-1), 8)])
