"""-m gpu: the fused L1 + SSIM HIP kernels (lg_loss_forward / lg_loss_backward through lightgaussian_amd.loss_utils)
against (1) golden vectors produced by the reference's own utils/loss_utils.py and (2) the float64 numpy oracle.
Tolerance: 1e-4 relative (north_star's floating-point bar); observed agreement is ~1e-6."""
import os

import numpy as np
import pytest
import torch

from lightgaussian_amd import loss_utils as LU
from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu
TOL = 1e-4
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_loss.npz"))
CASES = sorted({k.split(".")[0] for k in G.files if "." in k})
DEV = "cuda:0"


def _run(x, y, lam):
    xt = torch.tensor(x, device=DEV, requires_grad=True)
    yt = torch.tensor(y, device=DEV)
    Ll1 = LU.l1_loss(xt, yt)                       # the reference's call pattern (prune_finetune.py:161-164)
    loss = (1.0 - lam) * Ll1 + lam * (1.0 - LU.ssim(xt, yt))
    loss.backward()
    return float(Ll1.detach()), float(loss.detach()), xt.grad.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("name", CASES)
def test_hip_loss_matches_reference_golden(name):
    x, y, lam = G[f"{name}.x"], G[f"{name}.y"], float(G["lambda"])
    l1, loss, g = _run(x, y, lam)
    assert l1 == pytest.approx(float(G[f"{name}.l1"]), rel=TOL, abs=1e-8)
    assert loss == pytest.approx(float(G[f"{name}.loss"]), rel=TOL)
    ref = G[f"{name}.grad"].astype(np.float64)
    assert np.abs(g - ref).max() <= TOL * max(np.abs(ref).max(), 1.0 / x.size)


@pytest.mark.parametrize("shp,seed", [((3, 270, 480), 1), ((3, 33, 31), 2), ((1, 64, 64), 3), ((2, 3, 40, 50), 4), ((3, 1080, 1920), 5)])
def test_hip_loss_matches_oracle(shp, seed):
    rng = np.random.default_rng(seed)
    x = rng.random(shp, dtype=np.float32)
    y = np.clip(x + 0.1 * rng.standard_normal(shp).astype(np.float32), 0, 1).astype(np.float32)
    lam = 0.2
    l1, loss, g = _run(x, y, lam)
    flat = lambda a: a.reshape((-1,) + a.shape[-2:])   # conv2d groups: every (batch, channel) plane is independent
    assert l1 == pytest.approx(LO.l1_loss(x, y), rel=TOL)
    assert loss == pytest.approx(LO.l1_dssim(flat(x), flat(y), lam), rel=TOL)
    ref = LO.l1_dssim_grad(flat(x), flat(y), lam).reshape(x.shape)
    assert np.abs(g - ref).max() <= TOL * np.abs(ref).max()


def test_one_launch_serves_l1_and_ssim_and_is_deterministic():
    from lightgaussian_amd import _lib, rasterizer
    rng = np.random.default_rng(7)
    x = rng.random((3, 100, 130), dtype=np.float32); y = rng.random((3, 100, 130), dtype=np.float32)
    rasterizer.set_option("profile", True)
    try:
        _lib.profile_reset()
        a = _run(x, y, 0.2)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        assert prof["loss_fwd"][1] == 1 and prof["loss_bwd"][1] == 1   # l1_loss + ssim = one forward, one backward launch
    finally:
        rasterizer.set_option("profile", False)
        _lib.profile_reset()
    b = _run(x, y, 0.2)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])   # fixed-order block sums: run-to-run identical


def test_identical_images_and_separate_terms():
    x = torch.rand(3, 50, 70, device=DEV)
    assert float(LU.ssim(x, x.clone())) == pytest.approx(1.0, abs=1e-6)
    assert float(LU.l1_loss(x, x.clone())) == 0.0
    # gradient of ssim alone / l1 alone (the other output of the node gets no gradient)
    xr = x.clone().requires_grad_(True); y = torch.rand(3, 50, 70, device=DEV)
    LU.ssim(xr, y).backward()
    ref = LO.ssim_grad(x.cpu().numpy(), y.cpu().numpy())
    assert np.abs(xr.grad.cpu().numpy() - ref).max() <= TOL * np.abs(ref).max()
    xr2 = x.clone().requires_grad_(True)
    LU.l1_loss(xr2, y).backward()
    assert np.array_equal(xr2.grad.cpu().numpy(), (np.sign(x.cpu().numpy() - y.cpu().numpy()) / x.numel()).astype(np.float32))
    loss, l1 = LU.l1_dssim_loss(x, y, 0.2)
    assert float(loss) == pytest.approx(LO.l1_dssim(x.cpu().numpy(), y.cpu().numpy(), 0.2), rel=TOL)


@pytest.mark.parametrize("shp", [(3, 37, 53), (3, 1080, 1920), (1, 5, 7)])
def test_l1_only_kernel(shp):
    """LG_FLAG_L1_ONLY: the streaming L1 kernels give the same value and the same (exact) sign gradient as the fused ones."""
    rng = np.random.default_rng(11)
    x = rng.random(shp, dtype=np.float32); y = rng.random(shp, dtype=np.float32)
    y.flat[::7] = x.flat[::7]                                              # exact ties: gradient 0
    xt = torch.tensor(x, device=DEV, requires_grad=True); yt = torch.tensor(y, device=DEV)
    l = LU.l1_loss_only(xt, yt)
    (3.0 * l).backward()
    assert float(l.detach()) == pytest.approx(LO.l1_loss(x, y), rel=1e-6)
    assert np.array_equal(xt.grad.cpu().numpy(), (3.0 * np.sign(x - y) / x.size).astype(np.float32))


def test_loss_rejects_cpu_tensors_and_mismatched_shapes():
    with pytest.raises(RuntimeError):
        LU.l1_loss(torch.rand(3, 8, 8), torch.rand(3, 8, 8))
    x = torch.rand(3, 8, 8, device=DEV)
    assert float(LU.ssim(x, x, window_size=7)) == pytest.approx(1.0, abs=1e-5)     # (accepted since round 2, as the reference does)
    with pytest.raises(ValueError):
        LU.l1_loss(x, torch.rand(3, 8, 9, device=DEV))


def test_loss_gradient_reaches_the_gaussians_through_render():
    from lightgaussian_amd.gaussian_renderer import render
    from lightgaussian_amd.synthetic import make_gaussians, orbit_camera, PipelineParams
    pc = make_gaussians(5000, sh_degree=3).to(DEV).requires_grad_(True)
    cam = orbit_camera(0, 8, 160, 96).to(DEV)
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        gt = render(orbit_camera(1, 8, 160, 96).to(DEV), pc, PipelineParams(), bg)["render"].clone()
    image = render(cam, pc, PipelineParams(), bg)["render"]
    Ll1 = LU.l1_loss(image, gt)
    loss = 0.8 * Ll1 + 0.2 * (1.0 - LU.ssim(image, gt))
    loss.backward()
    g = pc._xyz.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_repeated_backward_recomputes_and_general_ssim_arguments():
    """l1_loss(x, y).backward() twice in a row must work (the memo is void once its graph ran backward), and the
    argument combinations the trainers never use (another window, per-image means) follow the reference formula."""
    x = torch.rand(3, 40, 60, device=DEV, requires_grad=True); y = torch.rand(3, 40, 60, device=DEV)
    LU.l1_loss(x, y).backward()
    g1 = x.grad.clone(); x.grad = None
    LU.l1_loss(x, y).backward()
    assert torch.equal(g1, x.grad)
    a = LU.l1_loss(x, y); b = LU.ssim(x, y)                       # the reference's pair: one node, then the memo is dropped
    assert LU._memo.last is None
    (a + b).backward()
    xb = torch.rand(2, 3, 40, 60, device=DEV); yb = torch.rand(2, 3, 40, 60, device=DEV)
    per_image = LU.ssim(xb, yb, size_average=False)
    assert per_image.shape == (2,)
    for i in range(2):
        assert float(per_image[i]) == pytest.approx(float(LU.ssim(xb[i], yb[i])), rel=1e-5)
    assert float(LU.ssim(xb[0], yb[0], window_size=7)) == pytest.approx(LO.ssim(xb[0].cpu().numpy(), yb[0].cpu().numpy()), abs=0.05)


def test_lazy_loss_scalars_give_the_eager_gradient_bit_for_bit():
    """loss_utils.set_lazy (run.py --lazy-loss): the reference's two calls and its formula on LazyLoss scalars -- one fused forward,
    one fused backward, dL/dimage bit-identical to the eager formula, item() from the pinned copy equal to the eager value."""
    from lightgaussian_amd import _lib, rasterizer
    rng = np.random.default_rng(21)
    x = rng.random((3, 120, 200), dtype=np.float32); y = rng.random((3, 120, 200), dtype=np.float32)
    lam = 0.2

    def run(lazy):
        xt = torch.tensor(x, device=DEV, requires_grad=True); yt = torch.tensor(y, device=DEV)
        prev = LU.set_lazy(lazy)
        try:
            Ll1 = LU.l1_loss(xt, yt)
            loss = (1.0 - lam) * Ll1 + lam * (1.0 - LU.ssim(xt, yt))
        finally:
            LU.set_lazy(prev)
        assert isinstance(loss, LU.LazyLoss) == lazy and isinstance(Ll1, LU.LazyLoss) == lazy
        loss.backward()
        return loss.item(), Ll1.item(), xt.grad.clone()

    rasterizer.set_option("profile", True)
    try:
        _lib.profile_reset()
        lazy = run(True)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        assert prof["loss_fwd"][1] == 1 and prof["loss_bwd"][1] == 1
    finally:
        rasterizer.set_option("profile", False)
        _lib.profile_reset()
    eager = run(False)
    assert torch.equal(lazy[2], eager[2])
    assert lazy[0] == pytest.approx(eager[0], rel=1e-6) and lazy[1] == pytest.approx(eager[1], rel=1e-6)
    # under no_grad (evaluation) and with a tensor operand: falls back to real tensors, same numbers
    prev = LU.set_lazy(True)
    try:
        with torch.no_grad():
            xt = torch.tensor(x, device=DEV); yt = torch.tensor(y, device=DEV)
            v = LU.l1_loss(xt, yt)
            assert isinstance(v, LU.LazyLoss) and not v.requires_grad
            assert float(v.mean().double()) == pytest.approx(eager[1], rel=1e-6)       # training_report's use (train_densify_prune.py)
            assert float(v * torch.tensor(2.0, device=DEV)) == pytest.approx(2 * eager[1], rel=1e-6)
    finally:
        LU.set_lazy(prev)


def test_event_timing_switch_of_the_runner():
    """run.py --lazy-loss / --no-iter-timing: the trainers' iter_start.elapsed_time(iter_end) on a pair that has not completed
    raises in torch; 'wait' waits for the end event, 'skip' returns NaN, None restores torch's method."""
    import math
    from lightgaussian_amd import run as lg_run
    a = torch.randn(4096, 4096, device=DEV)
    orig = torch.cuda.Event.elapsed_time

    def pair():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(30):
            a @ a
        e.record()
        return s, e

    s, e = pair()
    if not e.query():                                   # (a very fast device could have finished already)
        with pytest.raises(RuntimeError):
            s.elapsed_time(e)
    try:
        lg_run.event_timing("wait")
        s, e = pair()
        assert s.elapsed_time(e) > 0.0 and e.query()
        lg_run.event_timing("skip")
        s, e = pair()
        v = s.elapsed_time(e)
        assert math.isnan(v) or v > 0.0
        torch.cuda.synchronize()
        assert s.elapsed_time(e) > 0.0                  # a finished pair is timed as usual
    finally:
        lg_run.event_timing(None)
    assert torch.cuda.Event.elapsed_time is orig


@pytest.mark.parametrize("shp", [(1, 1, 1), (1, 1, 5), (1, 5, 1), (1, 34, 64), (1, 35, 65), (1, 68, 128), (1, 69, 129), (2, 10, 63), (1, 44, 74),
                                 (1, 33, 10), (3, 7, 193), (1, 103, 9)])
def test_strip_kernel_edges_against_the_oracle(shp):
    """Round 4's strip decomposition (a wave per 64 columns x 34 rows, 5-pixel halo, second pixel of lanes 0..9): images narrower
    than the halo, exactly one strip / segment, one past it, single rows and columns."""
    rng = np.random.default_rng(100 + shp[1] * 7 + shp[2])
    x = rng.random(shp, dtype=np.float32)
    y = np.clip(x + 0.2 * rng.standard_normal(shp).astype(np.float32), 0, 1).astype(np.float32)
    l1, loss, g = _run(x, y, 0.2)
    assert l1 == pytest.approx(LO.l1_loss(x, y), rel=TOL, abs=1e-8)
    assert loss == pytest.approx(LO.l1_dssim(x, y, 0.2), rel=TOL)
    ref = LO.l1_dssim_grad(x, y, 0.2)
    assert np.abs(g - ref).max() <= TOL * np.abs(ref).max()
