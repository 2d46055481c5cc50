"""CPU: the numpy restatement of the photometric loss (oracle/loss_oracle.py) against golden vectors produced by the
reference's own utils/loss_utils.py (tests/golden/make_golden_loss.py) -- this pins the loss oracle."""
import os
import re

import numpy as np
import pytest

from oracle import loss_oracle as LO

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_loss.npz"))
CASES = sorted({k.split(".")[0] for k in G.files if "." in k})


def test_window_is_the_reference_window_bit_for_bit():
    assert np.array_equal(LO.gaussian_window(), G["window"])
    assert np.array_equal(LO._window2d().astype(np.float32), G["window2d"])


def test_hip_kernel_constants_are_the_reference_window():
    src = open(os.path.join(HERE, "..", "lightgaussian_amd", "csrc", "lg_loss.h")).read()
    body = re.search(r"LG_SSIM_W\[11\]\s*=\s*\{(.*?)\};", src, flags=re.S).group(1)
    vals = np.array([float.fromhex(t.strip().rstrip("f")) for t in body.split(",")], dtype=np.float64)
    assert np.array_equal(vals.astype(np.float32), G["window"]) and np.array_equal(vals, G["window"].astype(np.float64))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_outputs_and_autograd(name):
    x, y, lam = G[f"{name}.x"], G[f"{name}.y"], float(G["lambda"])
    # reference values are float32 torch results; the oracle is float64
    assert LO.l1_loss(x, y) == pytest.approx(float(G[f"{name}.l1"]), rel=2e-6, abs=1e-8)
    assert LO.ssim(x, y) == pytest.approx(float(G[f"{name}.ssim"]), rel=5e-6)
    assert LO.l1_dssim(x, y, lam) == pytest.approx(float(G[f"{name}.loss"]), rel=5e-6)
    g = LO.l1_dssim_grad(x, y, lam)
    ref = G[f"{name}.grad"].astype(np.float64)
    # 1e-4 relative; the floor (1e-4 of the L1 gradient scale 1/n) covers the identical-image case, where the true
    # gradient is 0 and the reference's float32 autograd returns rounding noise of 2e-9
    assert np.abs(g - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1.0 / x.size)


def test_identical_images_give_ssim_one_and_zero_l1():
    x = G["one_1x11x11.x"]
    assert LO.ssim(x, x) == pytest.approx(1.0, abs=1e-12) and LO.l1_loss(x, x) == 0.0
    assert np.all(LO.l1_grad(x, x) == 0.0)


def test_ssim_gradient_matches_finite_differences():
    rng = np.random.default_rng(3)
    x = rng.random((2, 9, 13)); y = rng.random((2, 9, 13))
    g = LO.ssim_grad(x, y)
    for idx in [(0, 0, 0), (1, 4, 6), (0, 8, 12), (1, 0, 7)]:
        e = np.zeros_like(x); e[idx] = 1e-6
        fd = (LO.ssim(x + e, y) - LO.ssim(x - e, y)) / 2e-6
        assert fd == pytest.approx(g[idx], rel=1e-4, abs=1e-9)
