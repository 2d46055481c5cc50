#!/usr/bin/env python3
"""Nearest-code search at the reference's shapes (vectree/vectree.py: 8192-entry codebook, 27 / 48 feature dimensions, 8192-row
chunks over all Gaussians): lg_vq_nearest (f32 MFMA, fused argmin) next to the reference's torch formulation
(-torch.cdist(x, embed).argmax(-1), vectree/vq.py:265-266) on the same GPU.  Also times the compaction after a prune
(prune.compact_tensors vs 21 boolean-index kernels)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import vq, prune

dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for d in (27, 48):
    n, K = 1_200_000, 8192
    g = torch.Generator().manual_seed(d)
    embed = (torch.randn(K, d, generator=g) * 0.3).to(dev)
    x = (embed[torch.randint(0, K, (n,), generator=g).to(dev)] + 0.1 * torch.randn(n, d, generator=g).to(dev))
    ms_hip, ind = timed(lambda: vq.nearest_code(x, embed))

    def ref():
        out = []
        for i in range(0, n, 8192):                     # the reference's chunking (vectree.py:93-96)
            out.append((-torch.cdist(x[i:i + 8192].unsqueeze(0), embed.unsqueeze(0), p=2)).argmax(-1)[0])
        return torch.cat(out)
    ms_ref, ind_ref = timed(ref, reps=2)
    agree = float((ind == ind_ref).float().mean())
    flops = 2.0 * n * K * (d + 1)
    print(f"vq d={d}: n={n} K={K}  lg_vq_nearest {ms_hip:.2f} ms ({flops / ms_hip / 1e9:.1f} TFLOP/s f32)  torch cdist+argmax {ms_ref:.2f} ms  "
          f"index agreement {agree:.6f}")

N = 3_000_000
g = torch.Generator().manual_seed(1)
shapes = [(N, 3), (N, 1, 3), (N, 15, 3), (N, 1), (N, 3), (N, 4)]
ts = []
for s in shapes:
    for _ in range(3):                                    # parameter + two Adam moments
        ts.append(torch.randn(*s, generator=g).to(dev))
ts += [torch.rand(N, 1, generator=g).to(dev), torch.rand(N, 1, generator=g).to(dev), torch.rand(N, generator=g).to(dev)]
keep = (torch.rand(N, generator=g) > 0.66).to(dev)
ms_hip, outs = timed(lambda: prune.compact_tensors(ts, keep))
ms_ref, refs = timed(lambda: [t[keep] for t in ts])
print(f"compaction of 21 tensors at N={N} (keep {int(keep.sum())}): compact_tensors {ms_hip:.2f} ms  torch boolean indexing {ms_ref:.2f} ms  "
      f"equal {all(torch.equal(a, b) for a, b in zip(outs, refs))}")
