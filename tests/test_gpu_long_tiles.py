"""-m gpu: long per-tile lists.  Lists longer than the segment length S are processed by the backward as independent
(tile, segment) work items starting from checkpoints the forward leaves (lg_blend_fwd / lg_blend_bwd, DESIGN long-tile
robustness).  With S = 64 / 128 every test scene has multi-segment tiles: the image must not change at all (checkpoints are
extra outputs), the gradients must agree with the unsegmented replay to float rounding and with the float64 oracle within
the usual bound; a heavy-tailed scene (one dense pile of faint splats) exercises lists of thousands of entries."""
import math

import numpy as np
import pytest
import torch

import common
import gpu_common
from common import syn
from lightgaussian_amd import rasterizer
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore():
    yield
    rasterizer.set_option("segment_length", 0)
    rasterizer.set_option("long_tiles", "auto")


def _np(kw):
    return {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in kw.items()}


def _scene(N, W, H, scale, opm, seed, heavy=0.0, ext=(2, 1.2, 2)):
    g = syn.make_gaussians(N, seed=seed, log_scale_mean=math.log(scale), opacity_mean=opm, extent=ext)
    if heavy:
        syn.make_heavy_tailed(g, frac=heavy, radius=0.3, log_scale_mean=math.log(0.06), opacity_mean=-3.5)
    cam = syn.orbit_camera(1, 7, W, H, radius=5.0)
    return common.scene_kwargs(g, cam, W, H, deg=3, bg=(0.2, 0.1, 0.3), as_torch=True)


SCENES = [dict(N=3000, W=200, H=120, scale=0.05, opm=1.0, seed=2), dict(N=800, W=128, H=96, scale=0.3, opm=-2.0, seed=3),
          dict(N=6000, W=160, H=96, scale=0.06, opm=-3.0, seed=4), dict(N=20000, W=192, H=128, scale=0.01, opm=-1.0, seed=5, heavy=0.3)]


@pytest.mark.parametrize("sc", SCENES, ids=lambda s: f"N{s['N']}_{s['W']}x{s['H']}")
@pytest.mark.parametrize("S", [64, 128])
def test_segmented_backward_equals_the_unsegmented_one(sc, S):
    kw = _scene(**sc)
    gimg = np.random.RandomState(3).randn(3, sc["H"], sc["W"]).astype(np.float32)
    rasterizer.set_option("long_tiles", "serial")               # this test is about the BACKWARD's segments: one forward walk for both
    rasterizer.set_option("segment_length", 1 << 20)            # effectively unsegmented
    a = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    rasterizer.set_option("segment_length", S)
    b = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["radii"], b["radii"])
    ref64 = oracle.forward(dtype=np.float64, **_np(kw)); g64 = oracle.backward(ref64, gimg)
    ref32 = oracle.forward(**_np(kw)); g32 = oracle.backward(ref32, gimg)
    longest = int(np.diff(np.asarray(ref32.ranges).reshape(-1, 2), axis=1).max()) if hasattr(ref32, "ranges") else None
    for name in a["grads"]:
        r = g64[name]
        ea, eb = gpu_common.rel_err(a["grads"][name].reshape(r.shape), r), gpu_common.rel_err(b["grads"][name].reshape(r.shape), r)
        floor = gpu_common.rel_err(g32[name], r)
        assert eb <= max(1e-4, 3.0 * floor), f"{name}: segmented rel err {eb:.3e} (unsegmented {ea:.3e}, fp32 oracle floor {floor:.3e}, longest list {longest})"
        assert gpu_common.rel_err(b["grads"][name], a["grads"][name]) <= max(2e-5, 2.0 * floor), name


def test_count_render_and_canonical_backward_with_segments():
    sc = SCENES[0]
    kw = _scene(**sc)
    gimg = np.random.RandomState(4).randn(3, sc["H"], sc["W"]).astype(np.float32)
    ref = oracle.forward(count=True, **_np(kw))
    rasterizer.set_option("segment_length", 64)
    out = gpu_common.hip_forward_backward(kw, count=True)
    assert np.array_equal(out["count"], ref.count) and np.array_equal(out["color"].view(np.uint32), ref.color.view(np.uint32))
    rasterizer.set_option("fast_exp", False)
    try:
        b = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    finally:
        rasterizer.set_option("fast_exp", True)
    g64 = oracle.backward(oracle.forward(dtype=np.float64, **_np(kw)), gimg)
    g32 = oracle.backward(oracle.forward(**_np(kw)), gimg)
    for name, r in g64.items():
        if r is None:
            continue
        assert gpu_common.rel_err(b["grads"][name].reshape(r.shape), r) <= max(1e-4, 3.0 * gpu_common.rel_err(g32[name], r)), name


def test_heavy_tailed_scene_has_long_lists_and_runs_with_the_default_segment():
    from lightgaussian_amd.gaussian_renderer import render
    dev = torch.device("cuda:0")
    g = syn.make_gaussians(400_000, seed=9)
    syn.make_heavy_tailed(g, frac=0.08)
    pc = g.to(dev).requires_grad_(True)
    cam = syn.orbit_camera(0, 10, 960, 540).to(dev)
    pkg = render(cam, pc, syn.PipelineParams(), torch.zeros(3, device=dev))
    saved = pkg["render"].grad_fn.saved_tensors                # (..., radii, geom, binning, img)
    T = ((960 + 15) // 16) * ((540 + 15) // 16)
    ranges = saved[-2][: T * 8].view(torch.int32).view(T, 2).cpu().numpy()
    n = ranges[:, 1] - ranges[:, 0]
    pkg["render"].sum().backward()
    assert n.max() > 4 * 512, n.max()                         # several segments on the densest tiles (default S = 512)
    assert torch.isfinite(pc._xyz.grad).all() and float(pc._xyz.grad.abs().sum()) > 0


# ---- parallel long-tile forward (lg_blend_fwd_seg / _scan / _rewalk): segments walked independently, joined by a scan with an
# ---- exact re-walk where a pixel terminates
PAR_SCENES = SCENES + [dict(N=6000, W=160, H=96, scale=0.08, opm=-0.5, seed=6),      # semi-opaque, deep: pixels saturate inside later segments
                       dict(N=4000, W=130, H=70, scale=0.1, opm=2.5, seed=7)]        # opaque: saturation inside the first segment


@pytest.mark.parametrize("sc", PAR_SCENES, ids=lambda s: f"N{s['N']}_{s['W']}x{s['H']}_op{s['opm']}")
@pytest.mark.parametrize("S", [64, 128])
def test_parallel_long_tile_forward_matches_the_serial_walk(sc, S):
    """Same include / exclude decisions, transmittances as regrouped products: the image agrees with the serial walk to float
    rounding, the gradients (which start from the forward's checkpoints, final T and n_contrib) to 2e-5 of each other and
    within the usual bound of the float64 oracle (5e-5 of each other on the opaque scenes)."""
    kw = _scene(**sc)
    gimg = np.random.RandomState(8).randn(3, sc["H"], sc["W"]).astype(np.float32)
    rasterizer.set_option("segment_length", S)
    rasterizer.set_option("long_tiles", "serial")
    a = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    assert rasterizer.set_option("long_tiles", "parallel") == "serial"
    b = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    assert np.array_equal(a["radii"], b["radii"])
    assert np.abs(a["color"] - b["color"]).max() <= 3e-6, np.abs(a["color"] - b["color"]).max()
    ref64 = oracle.forward(dtype=np.float64, **_np(kw)); g64 = oracle.backward(ref64, gimg)
    ref32 = oracle.forward(**_np(kw)); g32 = oracle.backward(ref32, gimg)
    assert np.abs(b["color"] - ref32.color).max() <= 1e-5
    for name in a["grads"]:
        r = g64[name]
        floor = gpu_common.rel_err(g32[name], r)
        assert gpu_common.rel_err(b["grads"][name].reshape(r.shape), r) <= max(1e-4, 3.0 * floor), name
        # (opaque scenes amplify the last-bit differences of T through the backward's T / (1 - alpha) replay)
        assert gpu_common.rel_err(b["grads"][name], a["grads"][name]) <= max(5e-5, 4.0 * floor), name


def test_parallel_long_tile_forward_leaves_n_contrib_and_final_T_of_the_serial_walk():
    """The per-pixel outputs the backward starts from: n_contrib (index of the last contributor: exact -- terminations are
    resolved by the sequential re-walk) and final T (to float rounding), read from the saved image state."""
    from lightgaussian_amd.gaussian_renderer import render
    dev = torch.device("cuda:0")
    W, H = 192, 128
    g = syn.make_gaussians(20000, seed=5, log_scale_mean=math.log(0.03), opacity_mean=-0.5, extent=(2, 1.2, 2))
    syn.make_heavy_tailed(g, frac=0.3, radius=0.3, log_scale_mean=math.log(0.06), opacity_mean=-2.0)
    pc = g.to(dev)
    cam = syn.orbit_camera(1, 7, W, H, radius=5.0).to(dev)
    rasterizer.set_option("segment_length", 64)
    outs = {}
    for mode in ("serial", "parallel"):
        rasterizer.set_option("long_tiles", mode)
        pkg = render(cam, pc.requires_grad_(True), syn.PipelineParams(), torch.zeros(3, device=dev))
        img = pkg["render"].grad_fn.saved_tensors[-1]
        P = W * H
        off = ((P * 4 + 255) // 256) * 256
        outs[mode] = (pkg["render"].detach().clone(), img[: P * 4].view(torch.float32).clone(), img[off: off + P * 4].view(torch.int32).clone())
    assert int(outs["serial"][2].max()) > 4 * 64                 # lists of many segments were walked
    assert (outs["serial"][1] < 1e-3).float().mean() > 0.01        # and some pixels did saturate (the re-walk ran)
    assert torch.equal(outs["serial"][2], outs["parallel"][2])
    assert float((outs["serial"][1] - outs["parallel"][1]).abs().max()) <= 1e-6
    assert float((outs["serial"][0] - outs["parallel"][0]).abs().max()) <= 3e-6


def test_long_tile_mode_auto_is_a_pure_function_of_the_view():
    """ "auto" (the default): a list goes through the parallel kernels when it is longer than two segments and four times the
    view's mean list -- decided on the device from THIS view's instance count (lg_par_min).  No history: the first render of an
    outlier scene already takes the parallel walk, renders of other scenes in between change nothing, and two renders of the same
    inputs are bit-identical.  Ordinary multi-segment lists stay with the serial walk, so a scene without an outlier renders
    exactly as under "serial"."""
    # 27 000 tiny splats spread over 300 tiles (mean list ~ 100) + a pile of 3 000 larger faint ones on a handful of tiles
    g = syn.make_gaussians(30000, seed=5, log_scale_mean=math.log(0.01), opacity_mean=-1.0, extent=(2, 1.2, 2))
    syn.make_heavy_tailed(g, frac=0.1, radius=0.1, log_scale_mean=math.log(0.06), opacity_mean=-3.5)
    kw = common.scene_kwargs(g, syn.orbit_camera(1, 7, 320, 240, radius=5.0), 320, 240, deg=3, bg=(0.2, 0.1, 0.3), as_torch=True)
    small = _scene(**SCENES[0])
    gimg = np.random.RandomState(5).randn(3, 240, 320).astype(np.float32)
    rasterizer.set_option("segment_length", 64)
    rasterizer.set_option("long_tiles", "auto")
    first = gpu_common.hip_forward_backward(kw, grad_image=gimg)                     # nothing rendered before in this mode
    gpu_common.hip_forward_backward(small)                                           # another scene in between
    second = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    rasterizer.set_option("long_tiles", "serial")
    serial = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    assert np.array_equal(first["color"], second["color"])                           # history-free, run-to-run bit-identical
    for name in first["grads"]:
        assert np.array_equal(first["grads"][name], second["grads"][name]), name
    assert not np.array_equal(first["color"], serial["color"])                       # the outlier lists took the parallel walk ...
    assert np.abs(first["color"] - serial["color"]).max() <= 3e-6                    # ... same image to float rounding
    # a scene without an outlier list: "auto" == "serial", bit for bit
    c = gpu_common.hip_forward_backward(small)["color"]
    rasterizer.set_option("long_tiles", "auto")
    a = gpu_common.hip_forward_backward(small)["color"]
    assert np.array_equal(a, c)


def test_the_backward_runs_with_the_segment_length_of_its_forward():
    """The segment length is part of the call (lg_view.segment_length), snapshotted by the forward: changing the process default
    between a forward and its backward changes nothing (r2: a process-wide word read independently by both silently corrupted
    the checkpoint addressing).  At the C ABI a backward handed another segment length than its forward refuses to touch the
    buffers: zero gradients, and an error under LG_FLAG_DEBUG."""
    from lightgaussian_amd.gaussian_renderer import render
    dev = torch.device("cuda:0")
    g = syn.make_gaussians(6000, seed=4, log_scale_mean=math.log(0.06), opacity_mean=-3.0, extent=(2, 1.2, 2))
    cam = syn.orbit_camera(1, 7, 160, 96, radius=5.0).to(dev)
    pipe, bg = syn.PipelineParams(), torch.zeros(3, device=dev)
    gimg = torch.randn(3, 96, 160, device=dev)
    grads = []
    for flip in (False, True):
        pc = g.to(dev).requires_grad_(True)
        rasterizer.set_option("segment_length", 64)
        img = render(cam, pc, pipe, bg)["render"]
        if flip:
            rasterizer.set_option("segment_length", 128)          # between forward and backward
        (img * gimg).sum().backward()
        grads.append([p.grad.clone() for p in (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity)])
    for a, b in zip(*grads):
        assert torch.equal(a, b) and float(a.abs().sum()) > 0
    # C ABI: forward at S = 64, backward deliberately at S = 128 (the options snapshot of the autograd node overwritten)
    pc = g.to(dev).requires_grad_(True)
    rasterizer.set_option("segment_length", 64)
    img = render(cam, pc, pipe, bg)["render"]
    img.grad_fn.opts = dict(img.grad_fn.opts, segment_length=128)
    (img * gimg).sum().backward()
    assert float(pc._xyz.grad.abs().sum()) == 0.0 and float(pc._opacity.grad.abs().sum()) == 0.0
    pipe_dbg = syn.PipelineParams(); pipe_dbg.debug = True
    pc = g.to(dev).requires_grad_(True)
    img = render(cam, pc, pipe_dbg, bg)["render"]
    img.grad_fn.opts = dict(img.grad_fn.opts, segment_length=128)
    with pytest.raises(Exception, match="segment_length differs"):
        (img * gimg).sum().backward()
