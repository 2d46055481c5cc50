"""-m gpu: the two call shapes every reference training run produces and that no parity case exercised before round 4.

  (a) ACTIVE SH degree below the STORED degree.  GaussianModel keeps `_features_rest` at max_sh_degree (16 coefficients) and
      starts at active_sh_degree 0, stepping it up every 1000 iterations (scene/gaussian_model.py:46,125-127,
      prune_finetune.py:139, train_densify_prune.py:114); the rasterizer is called with shs [N,16,3] and sh_degree in {0,1,2}
      (gaussian_renderer/__init__.py:61,165).  Checked through GaussianRasterizer, render() literal and render() fused:
      forward / count / backward against the oracle, and the gradient beyond the active coefficients exactly zero.
  (b) scaling_modifier != 1 (gaussian_renderer/__init__.py:27,58,80; the GUI path of prune_finetune.py:111-115): scales +
      rotations, cov3D_precomp = get_covariance(modifier), and the fused raw-parameter path; image + every gradient.

Tolerances as in test_gpu_parity.py: integer outputs and the canonical image bit-identical; hardware-exp image <= 1e-4;
gradients <= max(1e-4, 3 x the float32 oracle's own error against the float64 oracle).
"""
import math

import numpy as np
import pytest
import torch

import common
import gpu_common
from common import syn
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
RAW = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _np(kw):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in kw.items()}


def _scene(N=2500, W=144, H=96, seed=21, scale=0.03, opm=0.0, active=3, stored=3):
    g = syn.make_gaussians(N, sh_degree=stored, seed=seed, log_scale_mean=math.log(scale), opacity_mean=opm, extent=(2, 1.2, 2),
                           log_scale_std=0.5, rest_std=0.15)
    g.active_sh_degree = active
    cam = syn.orbit_camera(2, 7, W, H, radius=5.0)
    return g, cam


def _check_grads(hip, g32, g64, what=""):
    for name, g in hip.items():
        r = g64[name]
        assert r is not None, name
        floor = gpu_common.rel_err(g32[name], r)
        err = gpu_common.rel_err(np.asarray(g).reshape(np.shape(r)), r)
        assert np.isfinite(g).all(), f"{what} grad {name} not finite"
        assert err <= max(TOL, 3.0 * floor), f"{what} grad {name}: rel err {err:.3e} (fp32 oracle floor {floor:.3e})"


# ---- (a) active degree below stored degree -------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [0, 1, 2])
def test_rasterizer_with_sixteen_stored_coefficients_and_a_lower_active_degree(D):
    """GaussianRasterizer(shs=[N,16,3], sh_degree=D): the reference's call while active_sh_degree is still climbing."""
    g, cam = _scene()
    W, H = 144, 96
    kw = common.scene_kwargs(g, cam, W, H, deg=3, bg=(0.2, 0.1, 0.3), as_torch=True)      # shs = all 16 stored coefficients
    kw["sh_degree"] = D
    assert kw["shs"].shape[1] == 16
    ref = oracle.forward(count=True, **_np(kw))
    # the stored-but-inactive coefficients must not leak into the colours: same image as with the row cut to (D + 1)^2
    cut = dict(_np(kw)); cut["shs"] = np.ascontiguousarray(cut["shs"][:, : (D + 1) ** 2])
    assert np.array_equal(oracle.forward(**cut).color, ref.color)
    out = gpu_common.hip_forward_backward(kw, count=True)
    assert np.array_equal(out["radii"], ref.radii)
    assert np.array_equal(out["count"], ref.count)
    assert np.array_equal(out["score"].view(np.uint32), ref.score.view(np.uint32))
    assert np.array_equal(out["color"].view(np.uint32), ref.color.view(np.uint32))
    gimg = np.random.RandomState(5 + D).randn(3, H, W).astype(np.float32)
    g32 = oracle.backward(ref, gimg)
    ref64 = oracle.forward(dtype=np.float64, **_np(kw)); g64 = oracle.backward(ref64, gimg)
    fast = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    assert gpu_common.rel_err(fast["color"], ref.color) <= TOL
    _check_grads(fast["grads"], g32, g64, f"D={D}")
    na = (D + 1) ** 2
    dsh = fast["grads"]["shs"]
    assert dsh.shape == (g.num, 16, 3)
    assert np.count_nonzero(dsh[:, na:]) == 0, "gradient beyond the active SH coefficients must be exactly zero"
    assert np.count_nonzero(dsh[:, :na]) > 0
    assert np.count_nonzero(g64["shs"][:, na:]) == 0


@pytest.mark.parametrize("fused", [False, True], ids=["literal", "fused"])
@pytest.mark.parametrize("D", [0, 1, 2])
def test_render_with_active_sh_degree_below_max_sh_degree(D, fused):
    """render() on a model with max_sh_degree 3 and active_sh_degree D (prune_finetune.py:139 before oneupSHdegree has run 3 times):
    literal getter pattern and getters fused into the kernels, against the oracle on the same activated inputs; gradients on the
    RAW parameters; `_features_rest.grad` beyond the active block exactly zero."""
    from lightgaussian_amd.gaussian_renderer import render
    from test_gpu_full_size import _activated_on_device, _oracle_kw, _chain_to_raw
    g, cam = _scene(active=D, stored=3)
    W, H = 144, 96
    dev = torch.device(DEV)
    pc = g.to(dev).requires_grad_(True)
    assert pc.active_sh_degree == D and pc.max_sh_degree == 3 and pc._features_rest.shape[1] == 15
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    gimg = np.random.RandomState(17 + D).randn(3, H, W).astype(np.float32)
    pkg = render(cam.to(dev), pc, syn.PipelineParams(), bg, options={"fuse_getters": fused})
    (pkg["render"] * torch.from_numpy(gimg).to(dev)).sum().backward()
    kw = _oracle_kw(_activated_on_device(pc), cam, W, H, D, bg.cpu().numpy())
    assert kw["shs"].shape[1] == 16
    f32 = oracle.forward(**kw); g32 = oracle.backward(f32, gimg)
    f64 = oracle.forward(dtype=np.float64, **kw); g64 = oracle.backward(f64, gimg)
    assert np.array_equal(pkg["radii"].cpu().numpy(), f32.radii)
    assert np.array_equal(pkg["visibility_filter"].cpu().numpy(), f32.radii > 0)
    assert gpu_common.rel_err(pkg["render"].detach().cpu().numpy(), f32.color) <= TOL
    hip_raw = {n: getattr(pc, n).grad.detach().cpu().numpy() for n in RAW}
    _check_grads(hip_raw, _chain_to_raw(g, g32), _chain_to_raw(g, g64), f"D={D} fused={fused}")
    nrest = (D + 1) ** 2 - 1
    assert np.count_nonzero(hip_raw["_features_rest"][:, nrest:]) == 0, "_features_rest.grad beyond the active block must be exactly zero"
    if D > 0:
        assert np.count_nonzero(hip_raw["_features_rest"][:, :nrest]) > 0
    # canonical arithmetic: image bit-identical to the oracle with the inactive coefficients present
    with torch.no_grad():
        ex = render(cam.to(dev), pc, syn.PipelineParams(), bg, options={"fuse_getters": fused, "fast_exp": False})["render"]
    assert np.array_equal(ex.cpu().numpy().view(np.uint32), f32.color.view(np.uint32))


@pytest.mark.parametrize("D", [0, 2])
def test_count_render_with_active_sh_degree_below_max_sh_degree(D):
    """The significance pass right after loading a checkpoint saved below degree 3 (prune.py:133-157 over count_render)."""
    from lightgaussian_amd.gaussian_renderer import count_render
    from test_gpu_full_size import _activated_on_device, _oracle_kw
    g, cam = _scene(active=D, stored=3, seed=23)
    W, H = 144, 96
    dev = torch.device(DEV)
    pc = g.to(dev)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        out = count_render(cam.to(dev), pc, syn.PipelineParams(), bg)
    ref = oracle.forward(count=True, **_oracle_kw(_activated_on_device(pc), cam, W, H, D, np.zeros(3)))
    assert np.array_equal(out["gaussians_count"].cpu().numpy(), ref.count)
    assert np.array_equal(out["important_score"].cpu().numpy().view(np.uint32), ref.score.view(np.uint32))
    assert np.array_equal(out["render"].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))


# ---- (b) scaling_modifier != 1 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mod", [0.5, 2.0])
def test_rasterizer_scale_modifier_with_scales_and_rotations(mod):
    g, cam = _scene(seed=31)
    W, H = 144, 96
    kw = common.scene_kwargs(g, cam, W, H, deg=3, bg=(0.1, 0.2, 0.3), as_torch=True)
    kw["scale_modifier"] = mod
    ref = oracle.forward(count=True, **_np(kw))
    base = dict(_np(kw)); base["scale_modifier"] = 1.0
    assert not np.array_equal(oracle.forward(**base).radii, ref.radii), "the modifier must change the footprint"
    out = gpu_common.hip_forward_backward(kw, count=True)
    assert np.array_equal(out["radii"], ref.radii)
    assert np.array_equal(out["count"], ref.count)
    assert np.array_equal(out["score"].view(np.uint32), ref.score.view(np.uint32))
    assert np.array_equal(out["color"].view(np.uint32), ref.color.view(np.uint32))
    gimg = np.random.RandomState(3).randn(3, H, W).astype(np.float32)
    g32 = oracle.backward(ref, gimg)
    ref64 = oracle.forward(dtype=np.float64, **_np(kw)); g64 = oracle.backward(ref64, gimg)
    fast = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    assert gpu_common.rel_err(fast["color"], ref.color) <= TOL
    _check_grads(fast["grads"], g32, g64, f"mod={mod}")
    # which convention "the scale gradient" follows (the published backward omits the modifier's own factor) is pinned on the CPU:
    # tests/test_oracle.py::test_scale_modifier_convention_of_the_oracle; here hip == oracle is what counts


@pytest.mark.parametrize("mod", [0.5, 2.0])
def test_rasterizer_scale_modifier_with_precomputed_covariance(mod):
    """pipe.compute_cov3D_python: cov3D_precomp = pc.get_covariance(scaling_modifier) (gaussian_renderer/__init__.py:80); the
    settings record still carries the modifier, which the rasterizer must then ignore."""
    g, cam = _scene(seed=32)
    W, H = 144, 96
    kw = common.scene_kwargs(g, cam, W, H, deg=3, bg=(0.1, 0.2, 0.3), as_torch=True)
    del kw["scales"], kw["rotations"]
    kw["cov3D_precomp"] = g.get_covariance(mod).contiguous()
    kw["scale_modifier"] = mod
    ref = oracle.forward(count=True, **_np(kw))
    # equal to rasterising the scales with the modifier, up to the rounding of the torch-side covariance
    sr = common.scene_kwargs(g, cam, W, H, deg=3, bg=(0.1, 0.2, 0.3)); sr["scale_modifier"] = mod
    assert gpu_common.rel_err(oracle.forward(**sr).color, ref.color) <= 1e-3
    out = gpu_common.hip_forward_backward(kw, count=True)
    assert np.array_equal(out["radii"], ref.radii)
    assert np.array_equal(out["count"], ref.count)
    assert np.array_equal(out["color"].view(np.uint32), ref.color.view(np.uint32))
    gimg = np.random.RandomState(4).randn(3, H, W).astype(np.float32)
    g32 = oracle.backward(ref, gimg)
    ref64 = oracle.forward(dtype=np.float64, **_np(kw)); g64 = oracle.backward(ref64, gimg)
    fast = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    assert gpu_common.rel_err(fast["color"], ref.color) <= TOL
    _check_grads(fast["grads"], g32, g64, f"precov mod={mod}")


@pytest.mark.parametrize("fused", [False, True], ids=["literal", "fused"])
@pytest.mark.parametrize("mod", [0.5, 2.0])
def test_render_scaling_modifier_reaches_the_raw_parameters(mod, fused):
    """render(..., scaling_modifier=mod) (prune_finetune.py:111-115) through the literal getters and through the fused raw path:
    image and every raw-parameter gradient against the oracle chained through the reference's getters."""
    from lightgaussian_amd.gaussian_renderer import render
    from test_gpu_full_size import _activated_on_device, _oracle_kw, _chain_to_raw
    g, cam = _scene(seed=33)
    W, H = 144, 96
    dev = torch.device(DEV)
    pc = g.to(dev).requires_grad_(True)
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    gimg = np.random.RandomState(9).randn(3, H, W).astype(np.float32)
    pkg = render(cam.to(dev), pc, syn.PipelineParams(), bg, mod, options={"fuse_getters": fused})
    (pkg["render"] * torch.from_numpy(gimg).to(dev)).sum().backward()
    kw = _oracle_kw(_activated_on_device(pc), cam, W, H, 3, bg.cpu().numpy())
    kw["scale_modifier"] = mod
    f32 = oracle.forward(**kw); g32 = oracle.backward(f32, gimg)
    f64 = oracle.forward(dtype=np.float64, **kw); g64 = oracle.backward(f64, gimg)
    assert np.array_equal(pkg["radii"].cpu().numpy(), f32.radii)
    assert gpu_common.rel_err(pkg["render"].detach().cpu().numpy(), f32.color) <= TOL
    hip_raw = {n: getattr(pc, n).grad.detach().cpu().numpy() for n in RAW}
    _check_grads(hip_raw, _chain_to_raw(g, g32), _chain_to_raw(g, g64), f"mod={mod} fused={fused}")
    with torch.no_grad():
        ex = render(cam.to(dev), pc, syn.PipelineParams(), bg, mod, options={"fuse_getters": fused, "fast_exp": False})["render"]
    assert np.array_equal(ex.cpu().numpy().view(np.uint32), f32.color.view(np.uint32))


@pytest.mark.parametrize("mod", [0.5, 2.0])
def test_render_scaling_modifier_with_python_covariance(mod):
    """pipe.compute_cov3D_python = True with a modifier: render() hands get_covariance(mod) to the rasterizer."""
    from lightgaussian_amd.gaussian_renderer import render
    g, cam = _scene(seed=34)
    W, H = 144, 96
    dev = torch.device(DEV)
    pc = g.to(dev)
    bg = torch.zeros(3, device=dev)
    pipe = syn.PipelineParams(compute_cov3D_python=True)
    with torch.no_grad():
        img = render(cam.to(dev), pc, pipe, bg, mod, options={"fast_exp": False})["render"]
        cov = pc.get_covariance(mod).contiguous()
    kw = common.scene_kwargs(g, cam, W, H, deg=3)
    del kw["scales"], kw["rotations"]
    kw["cov3D_precomp"] = cov.cpu().numpy()
    kw["means3D"] = pc.get_xyz.cpu().numpy(); kw["opacities"] = pc.get_opacity.cpu().numpy(); kw["shs"] = pc.get_features.contiguous().cpu().numpy()
    kw["scale_modifier"] = mod
    ref = oracle.forward(**kw)
    assert np.array_equal(img.cpu().numpy().view(np.uint32), ref.color.view(np.uint32))


# ---- round 4: the SH direction Jacobian saved by K1 (LG_FLAG_SAVE_SH_JACOBIAN) ---------------------------------------------------
@pytest.mark.parametrize("fused", [False, True], ids=["literal", "fused"])
@pytest.mark.parametrize("D", [0, 1, 2, 3])
def test_backward_with_the_saved_sh_jacobian_is_bit_identical_to_the_backward_that_reads_the_coefficients(D, fused, monkeypatch):
    """A differentiated forward makes K1 leave d rgb / d (view direction) per visible Gaussian (36 B) and K9 skip the SH coefficients
    (388 MB per view at C3).  Same operations in the same order: every gradient bit-identical to the path that re-reads the coefficients
    (option sh_jacobian=False, rounds 1-3), for every active degree, stored degree 3."""
    from lightgaussian_amd.gaussian_renderer import render
    g, cam = _scene(active=D, stored=3, seed=41)
    W, H = 144, 96
    dev = torch.device(DEV)
    bg = torch.tensor([0.1, 0.3, 0.2], device=dev)
    gimg = torch.from_numpy(np.random.RandomState(23).randn(3, H, W).astype(np.float32)).to(dev)
    res = []
    for off in (False, True):
        pc = g.to(dev).requires_grad_(True)
        pkg = render(cam.to(dev), pc, syn.PipelineParams(), bg, options={"fuse_getters": fused, "sh_jacobian": not off})
        (pkg["render"] * gimg).sum().backward()
        res.append([pkg["render"].detach().clone()] + [getattr(pc, n).grad.clone() for n in RAW])
    for n, a, b in zip(("image",) + RAW, *res):
        assert torch.equal(a, b), n
    assert float(res[0][1].abs().sum()) > 0 and (D == 0 or float(res[0][3].abs().sum()) > 0)
