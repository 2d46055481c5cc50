#!/usr/bin/env python3
"""CPU only: how many lanes of K7's wave a (tile, splat) instance occupies on the frozen C3 scene -- the number behind DESIGN 21.1's
verdict on the reduction variants the r3 review listed for lg_blend_bwd:
  (iii) "skip the LDS-transposed reduction when few lanes contributed": needs instances with popc(cmask) <= 4;
  (ii)  "reduce two entries per pass": needs the lane sets of consecutive entries to be small enough to share a pass.
A lane of K7 owns the four pixels (x % 8, y % 8) of its 16 x 16 tile; it is in cmask when alpha >= 1/255 on any of them (saturation
by earlier splats only removes lanes, so these are upper bounds on popc and lower bounds on the share of small sets).
Plain numpy restatement of the projection (EWA, +0.3 dilation, conic) for the splats whose centre falls into a window of tiles.
    python tools/lane_stats.py [--n 3000000] [--window 16]
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=3_000_000)
    ap.add_argument("--window", type=int, default=16, help="tiles per side of the sampled window around the image centre")
    a = ap.parse_args()
    W, H = 1920, 1080
    g = syn.make_gaussians(a.n)
    cam = syn.orbit_camera(0, 200, W, H)
    vm = cam.world_view_transform.numpy().astype(np.float64)        # row-vector convention
    pm = cam.full_proj_transform.numpy().astype(np.float64)
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    fx, fy = W / (2 * tanx), H / (2 * tany)
    xyz = g._xyz.numpy().astype(np.float64)
    ph = np.concatenate([xyz, np.ones((a.n, 1))], 1)
    v = ph @ vm
    hp = ph @ pm
    ndc = hp[:, :2] / (hp[:, 3:4] + 1e-7)
    ix = ((ndc[:, 0] + 1) * W - 1) * 0.5
    iy = ((ndc[:, 1] + 1) * H - 1) * 0.5
    x0, y0 = (W // 32 - a.window // 2) * 16, (H // 32 - a.window // 2) * 16
    x1, y1 = x0 + 16 * a.window, y0 + 16 * a.window
    sel = np.nonzero((v[:, 2] > 0.2) & (ix >= x0) & (ix < x1) & (iy >= y0) & (iy < y1))[0]
    s = np.exp(g._scaling.numpy().astype(np.float64)[sel])
    q = g._rotation.numpy().astype(np.float64)[sel]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = R * s[:, None, :]
    S3 = M @ M.transpose(0, 2, 1)
    t = v[sel, :3].copy()
    t[:, 0] = np.clip(t[:, 0] / t[:, 2], -1.3 * tanx, 1.3 * tanx) * t[:, 2]
    t[:, 1] = np.clip(t[:, 1] / t[:, 2], -1.3 * tany, 1.3 * tany) * t[:, 2]
    J = np.zeros((len(sel), 2, 3))
    J[:, 0, 0] = fx / t[:, 2]; J[:, 0, 2] = -fx * t[:, 0] / t[:, 2] ** 2
    J[:, 1, 1] = fy / t[:, 2]; J[:, 1, 2] = -fy * t[:, 1] / t[:, 2] ** 2
    Wm = vm[:3, :3].T                                              # world -> view rotation (column-vector form)
    T = J @ Wm
    c2 = T @ S3 @ T.transpose(0, 2, 1)
    ca, cb, cc = c2[:, 0, 0] + 0.3, c2[:, 0, 1], c2[:, 1, 1] + 0.3
    det = ca * cc - cb * cb
    A, B, C = cc / det, -cb / det, ca / det
    op = 1 / (1 + np.exp(-g._opacity.numpy().astype(np.float64)[sel, 0]))
    mid = 0.5 * (ca + cc)
    rad = np.ceil(3 * np.sqrt(mid + np.sqrt(np.maximum(0.1, mid * mid - det))))
    cx, cy = ix[sel], iy[sel]
    lanes, pixels, blocks = [], [], []
    py, px = np.mgrid[0:16, 0:16]
    for k in range(len(sel)):
        if op[k] < 1 / 255:
            continue
        tx0, tx1 = int(max(0, (cx[k] - rad[k]) // 16)), int(min((W + 15) // 16, (cx[k] + rad[k] + 15) // 16))
        ty0, ty1 = int(max(0, (cy[k] - rad[k]) // 16)), int(min((H + 15) // 16, (cy[k] + rad[k] + 15) // 16))
        for ty in range(ty0, ty1):
            for tx in range(tx0, tx1):
                dx = cx[k] - (tx * 16 + px); dy = cy[k] - (ty * 16 + py)
                power = -0.5 * (A[k] * dx * dx + C[k] * dy * dy) - B[k] * dx * dy
                hit = (power <= 0) & (np.minimum(0.99, op[k] * np.exp(power)) >= 1 / 255)
                n = int(hit.sum())
                if n == 0:
                    continue                                       # the exact footprint test removes these instances
                lane = hit[:8, :8] | hit[:8, 8:] | hit[8:, :8] | hit[8:, 8:]
                lanes.append(int(lane.sum())); pixels.append(n)
                blocks.append(int(hit[:8, :8].any()) + int(hit[:8, 8:].any()) + int(hit[8:, :8].any()) + int(hit[8:, 8:].any()))
    lanes, pixels, blocks = np.array(lanes), np.array(pixels), np.array(blocks)
    print(f"{len(sel)} splats with their centre in the {a.window} x {a.window} tile window, {len(lanes)} (tile, splat) instances with a pixel at alpha >= 1/255")
    print(f"pixels per instance: mean {pixels.mean():.1f}, median {np.median(pixels):.0f}; 8x8 blocks hit per instance: mean {blocks.mean():.2f}")
    print(f"lanes of the wave in cmask (upper bound): mean {lanes.mean():.1f}, median {np.median(lanes):.0f}")
    for b in (1, 2, 4, 8, 16, 32):
        print(f"  instances with <= {b:2d} lanes: {100.0 * (lanes <= b).mean():5.1f} %")
    share = np.array([(lanes[blocks == nb]).mean() if (blocks == nb).any() else 0 for nb in (1, 2, 3, 4)])
    print("  mean lanes by blocks hit (1, 2, 3, 4):", np.round(share, 1), " share of instances:", np.round([(blocks == nb).mean() for nb in (1, 2, 3, 4)], 3))
    print(f"useful lanes per evaluated (block, instance) pair: {pixels.sum() / (64.0 * blocks.sum()):.3f}")


if __name__ == "__main__":
    main()
