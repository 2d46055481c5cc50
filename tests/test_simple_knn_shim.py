"""simple_knn._C.distCUDA2 import shim (off the hot path; lets the reference's scene/gaussian_model.py import)."""
import numpy as np
import torch

from simple_knn._C import distCUDA2


def test_matches_brute_force():
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(700, 3, generator=g)
    pts[10] = pts[11] + 1e-4                      # near-duplicate pair
    got = distCUDA2(pts, chunk=256).numpy()
    p = pts.double().numpy()
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    want = np.sort(d2, axis=1)[:, :3].mean(1)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-9)
    assert got.dtype == np.float32 and got.shape == (700,)


def test_reference_call_pattern():
    pts = torch.rand(50, 3)
    dist2 = torch.clamp_min(distCUDA2(pts.float()), 0.0000001)   # scene/gaussian_model.py:152-153
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
    assert torch.isfinite(scales).all()
