"""How many (block, splat) evaluations do 8x8, 8x4, 4x4 blocks need on the benchmark scene's splat statistics? (CPU, oracle)"""
import math, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from common import syn
from oracle import oracle
N, W, H = 300_000, 1920, 1080
g = syn.make_gaussians(N)
cam = syn.orbit_camera(0, 200, W, H)
kw = common.scene_kwargs(g, cam, W, H)
f = oracle.forward(**kw)
vis = f.radii > 0
xy = f.saved["xy"][vis].astype(np.float64); co = f.saved["conic_opacity"][vis].astype(np.float64)
A, B, C, op = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
keep = op >= 1 / 255.0
xy, A, B, C, op = xy[keep], A[keep], B[keep], C[keep], op[keep]
n = len(op)
print("visible", vis.sum(), "with opacity >= 1/255", n)
# brute force: for each splat, pixels with alpha >= 1/255 in a 41x41 window around it
R = 20
ox = np.floor(xy[:, 0]).astype(int); oy = np.floor(xy[:, 1]).astype(int)
gx = np.arange(-R, R + 1)
px = ox[:, None] + gx[None, :]; py = oy[:, None] + gx[None, :]
dx = xy[:, 0][:, None, None] - px[:, None, :]; dy = xy[:, 1][:, None, None] - py[:, :, None]
power = -0.5 * (A[:, None, None] * dx * dx + C[:, None, None] * dy * dy) - B[:, None, None] * dx * dy
alpha = np.minimum(0.99, op[:, None, None] * np.exp(np.minimum(power, 0)))
hit = (power <= 0) & (alpha >= 1 / 255.0) & (px[:, None, :] >= 0) & (px[:, None, :] < W) & (py[:, :, None] >= 0) & (py[:, :, None] < H)
print("contributing pixels per splat: mean", hit.sum((1, 2)).mean(), "clipped window?", hit[:, 0, :].any() or hit[:, :, 0].any())
PX = np.broadcast_to(px[:, None, :], hit.shape); PY = np.broadcast_to(py[:, :, None], hit.shape)
def count_blocks(bw, bh):
    bid = (PY // bh) * 100000 + (PX // bw)
    tot = 0
    for i in range(0, n, 20000):
        h = hit[i:i + 20000]; b = np.where(h, bid[i:i + 20000], -1).reshape(h.shape[0], -1)
        b.sort(axis=1)
        tot += ((b[:, 1:] != b[:, :-1]) & (b[:, 1:] >= 0)).sum() + (b[:, 0] >= 0).sum()
    return tot / n
for bw, bh in ((16, 16), (8, 8), (8, 4), (4, 4), (4, 2), (2, 2)):
    c = count_blocks(bw, bh)
    print(f"blocks {bw}x{bh}: {c:.3f} per splat (blocks holding >= 1 contributing pixel); lanes evaluated per splat {c * bw * bh:.1f}; useful fraction {hit.sum((1,2)).mean() / (c * bw * bh):.3f}")
