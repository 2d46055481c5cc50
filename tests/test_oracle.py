"""The CPU oracle against an independent autograd derivation + invariants + hand-checkable scenes.  No GPU."""
import math

import numpy as np
import pytest
import torch

import common
from common import syn
from oracle import oracle, torch_dense


def _dense_case(N, W, H, seed, precolor=False, precov=False, deg=3, big=False, opm=None):
    g = syn.make_gaussians(N, sh_degree=3, seed=seed, extent=(1.5, 1.0, 1.5), log_scale_mean=math.log(0.15 if big else 0.05),
                           opacity_mean=(opm if opm is not None else (1.0 if big else 0.0)))
    cam = syn.orbit_camera(1, 7, W, H, radius=4.0)
    dd = torch.float64
    t = dict(means3D=g.get_xyz.to(dd).requires_grad_(), opacities=g.get_opacity.to(dd).detach().requires_grad_(),
             W=W, H=H, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.tensor([0.2, 0.5, 0.9], dtype=dd),
             viewmatrix=cam.world_view_transform.to(dd), projmatrix=cam.full_proj_transform.to(dd), campos=cam.camera_center.to(dd),
             sh_degree=deg)
    M = (deg + 1) ** 2
    if precolor:
        t["colors_precomp"] = torch.rand(N, 3, dtype=dd, generator=torch.Generator().manual_seed(seed)).requires_grad_()
    else:
        t["shs"] = g.get_features.to(dd)[:, :M].detach().clone().requires_grad_()
    if precov:
        t["cov3D_precomp"] = g.get_covariance().to(dd).detach().requires_grad_()
    else:
        t["scales"] = g.get_scaling.to(dd).detach().requires_grad_()
        t["rotations"] = g.get_rotation.to(dd).detach().requires_grad_()
    return t


@pytest.mark.parametrize("cfg", [dict(N=40, W=48, H=40, seed=1), dict(N=60, W=64, H=33, seed=2, precolor=True),
                                 dict(N=60, W=40, H=40, seed=3, precov=True, deg=2), dict(N=80, W=40, H=48, seed=4, deg=1, big=True),
                                 dict(N=30, W=32, H=32, seed=5, deg=0, big=True), dict(N=150, W=32, H=32, seed=6, big=True, opm=3.0)],
                         ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_oracle_f64_matches_autograd_twin(cfg):
    t = _dense_case(**cfg)
    N, W, H = cfg["N"], cfg["W"], cfg["H"]
    means2D = torch.zeros(N, 3, dtype=torch.float64, requires_grad=True)
    color, radii, count = torch_dense.render_dense(means2D=means2D, **t)
    gimg = torch.randn(3, H, W, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
    (color * gimg).sum().backward()
    kwn = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in t.items()}
    f = oracle.forward(count=True, dtype=np.float64, **kwn)
    gr = oracle.backward(f, gimg.numpy())
    assert np.abs(f.color - color.detach().numpy()).max() < 1e-12
    assert np.array_equal(f.radii, radii.numpy()) and np.array_equal(f.count, count.numpy())
    if cfg.get("opm") == 3.0:
        assert f.saved["final_T"].min() < 2e-4, "this case is meant to exercise early termination"
    grads = dict(t, means2D=means2D)
    for name, g in gr.items():
        if g is None:
            continue
        ref = grads[name].grad.numpy().reshape(g.shape)
        err = np.abs(g - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < 1e-4, f"{name}: {err}"   # 1e-7 regulariser of the published cov2D backward <= 1.3e-5


@pytest.mark.parametrize("D", [0, 1, 2])
def test_stored_degree_three_with_a_lower_active_degree_matches_the_autograd_twin(D):
    """shs [N,16,3] with sh_degree D < 3 (scene/gaussian_model.py:46,125-127: every run starts at active_sh_degree 0 over 16 stored
    coefficients): colours use the first (D + 1)^2 coefficients only, the gradient of the rest is exactly zero."""
    t = _dense_case(50, 40, 36, seed=10 + D, deg=3)
    t["sh_degree"] = D
    N, W, H = 50, 40, 36
    assert t["shs"].shape[1] == 16
    means2D = torch.zeros(N, 3, dtype=torch.float64, requires_grad=True)
    color, radii, count = torch_dense.render_dense(means2D=means2D, **t)
    gimg = torch.randn(3, H, W, dtype=torch.float64, generator=torch.Generator().manual_seed(8))
    (color * gimg).sum().backward()
    kwn = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in t.items()}
    f = oracle.forward(count=True, dtype=np.float64, **kwn)
    gr = oracle.backward(f, gimg.numpy())
    assert np.abs(f.color - color.detach().numpy()).max() < 1e-12
    cut = dict(kwn); cut["shs"] = np.ascontiguousarray(kwn["shs"][:, : (D + 1) ** 2])
    assert np.array_equal(oracle.forward(dtype=np.float64, **cut).color, f.color)
    ref = t["shs"].grad.numpy()
    assert np.abs(gr["shs"] - ref).max() <= 1e-4 * np.abs(ref).max()
    assert np.count_nonzero(gr["shs"][:, (D + 1) ** 2:]) == 0 and np.count_nonzero(ref[:, (D + 1) ** 2:]) == 0
    for name in ("means3D", "opacities", "scales", "rotations"):
        r = t[name].grad.numpy().reshape(gr[name].shape)
        assert np.abs(gr[name] - r).max() <= 1e-4 * np.abs(r).max(), name


@pytest.mark.parametrize("mod", [0.5, 2.0])
def test_scale_modifier_convention_of_the_oracle(mod):
    """scale_modifier (gaussian_renderer/__init__.py:58; GUI path prune_finetune.py:111-115): the forward uses mod * scale; the
    published backward returns dL/d(mod * scale) as "the scale gradient" -- the modifier's own factor is omitted [RECALLED-UPSTREAM,
    SURVEY App. A].  Pinned here: rendering (scales, mod) equals rendering (mod * scales, 1) bit for bit (mod a power of two), every
    gradient included, and the latter's gradients equal the independent autograd derivation."""
    g = syn.make_gaussians(300, seed=3, extent=(1.5, 1.0, 1.5), log_scale_mean=math.log(0.06), opacity_mean=0.0)
    cam = syn.orbit_camera(1, 7, 64, 48, radius=4.0)
    kw = common.scene_kwargs(g, cam, 64, 48, bg=(0.1, 0.2, 0.3))
    gimg = np.random.RandomState(1).randn(3, 48, 64)
    for dt in (np.float32, np.float64):
        a = oracle.forward(dtype=dt, scale_modifier=mod, **kw)
        kb = dict(kw); kb["scales"] = (kw["scales"].astype(dt) * dt(mod))
        b = oracle.forward(dtype=dt, **kb)
        assert np.array_equal(a.color, b.color) and np.array_equal(a.radii, b.radii)
        ga, gb = oracle.backward(a, gimg), oracle.backward(b, gimg)
        for name in ("means2D", "means3D", "opacities", "shs", "scales", "rotations"):
            assert np.array_equal(ga[name], gb[name]), name
    # ... and the precomputed-covariance input ignores the modifier altogether
    kc = dict(kw); del kc["scales"], kc["rotations"]
    kc["cov3D_precomp"] = oracle.cov3d(kw["scales"], kw["rotations"], mod)
    assert np.array_equal(oracle.forward(scale_modifier=mod, **kc).color, oracle.forward(scale_modifier=1.0, **kc).color)
    assert np.array_equal(oracle.forward(**kc).color, oracle.forward(scale_modifier=mod, **kw).color)


def test_oracle_f32_close_to_f64_and_counts_equal():
    g = syn.make_gaussians(5000, seed=11, log_scale_mean=math.log(0.02))
    cam = syn.orbit_camera(2, 9, 200, 150)
    kw = common.scene_kwargs(g, cam, 200, 150, bg=(0.3, 0.3, 0.3))
    a = oracle.forward(count=True, **kw)
    b = oracle.forward(count=True, dtype=np.float64, **kw)
    assert np.abs(a.color - b.color).max() < 1e-4
    assert np.array_equal(a.radii, b.radii)
    assert np.count_nonzero(a.count != b.count) <= 5  # borderline threshold pairs only


def test_invariants_weights_sum_and_score_definition():
    g = syn.make_gaussians(3000, seed=5, log_scale_mean=math.log(0.03), opacity_mean=1.0)
    cam = syn.orbit_camera(0, 3, 128, 128)
    kw = common.scene_kwargs(g, cam, 128, 128, precolor=torch.ones(3000, 3), bg=(1.0, 1.0, 1.0))
    f = oracle.forward(count=True, **kw)
    assert np.allclose(f.color, 1.0, atol=2e-5)                      # sum(w) + T_final = 1
    op = kw["opacities"].reshape(-1)
    for i in np.random.RandomState(0).choice(3000, 300, replace=False):
        assert np.float32(f.score[i]) == np.float32(oracle.seqsum(op[i], int(f.count[i])))
    one = oracle.forward(count=True, weight_policy=oracle.W_ONE, **kw)
    assert np.array_equal(one.score, one.count.astype(np.float32))
    # count = number of (pixel, Gaussian) pairs that pass the three tests: recount through the ALPHA_T policy
    at = oracle.forward(count=True, weight_policy=oracle.W_ALPHA_T, **kw)
    assert np.array_equal(at.count, f.count)
    assert np.isclose(at.score.sum(), (1.0 - f.saved["final_T"]).sum(), rtol=1e-4)   # sum of blend weights


def _single(xyz, scale, opacity, W=64, H=64, rgb=(1.0, 0.5, 0.25), bg=(0.0, 0.0, 0.0)):
    cam = syn.orbit_camera(0, 4, W, H, radius=5.0)
    n = len(xyz)
    rot = np.tile(np.array([[1.0, 0, 0, 0]], np.float32), (n, 1))
    kw = dict(means3D=np.asarray(xyz, np.float32), opacities=np.asarray(opacity, np.float32).reshape(n, 1), W=W, H=H,
              tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=np.asarray(bg, np.float32),
              viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
              campos=cam.camera_center.numpy(), sh_degree=0, colors_precomp=np.tile(np.asarray(rgb, np.float32), (n, 1)),
              scales=np.asarray(scale, np.float32).reshape(n, 3), rotations=rot)
    return kw, cam


def test_known_answer_single_gaussian_centre_pixel():
    """One isotropic Gaussian on the optical axis: at its centre pixel alpha = min(0.99, opacity * exp(power)),
    colour = rgb * alpha; everything follows from the projection formulas by hand."""
    kw, cam = _single([[0.0, 0.0, 0.0]], [[0.1, 0.1, 0.1]], [0.8])
    f = oracle.forward(count=True, dtype=np.float64, **kw)
    assert f.radii[0] > 0
    x, y = f.saved["xy"][0]
    assert abs(x - 31.5) < 1e-6 and abs(y - 31.5) < 1e-6            # ((0+1)*64-1)/2
    fx = 64 / (2 * math.tan(cam.FoVx * 0.5))
    var = (0.1 * fx / 5.0) ** 2 + 0.3                              # EWA: sigma^2 * (f/z)^2 + low-pass
    A, B, Cc, op = f.saved["conic_opacity"][0]
    assert np.isclose(A, 1 / var, rtol=1e-6) and abs(B) < 1e-9 and np.isclose(Cc, 1 / var, rtol=1e-6)
    # eigenvalue with the published max(0.1, mid^2 - det) floor: lambda = var + sqrt(0.1) for an isotropic splat
    assert f.radii[0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    d2 = 0.5 ** 2 + 0.5 ** 2                                       # pixel (31,31) is half a pixel away in x and y
    alpha = 0.8 * math.exp(-0.5 * d2 / var)
    assert np.allclose(f.color[:, 31, 31], np.array([1.0, 0.5, 0.25]) * alpha, rtol=1e-6)  # camera matrices are fp32
    assert f.count[0] == int(((f.color[0] > 0)).sum())


def test_known_answer_depth_order_and_near_cull():
    kw, _ = _single([[0, 0, 1.0], [0, 0, -1.0], [0, 0, -6.0]], [[0.2] * 3, [0.2] * 3, [0.2] * 3], [0.5, 0.5, 0.5])
    kw["colors_precomp"] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    f = oracle.forward(dtype=np.float64, **kw)
    # camera sits at z=-5 looking at +z: the z=-1 Gaussian is in front of z=+1; z=-6 is behind the camera (culled)
    assert f.radii[2] == 0 and f.radii[0] > 0 and f.radii[1] > 0
    px = f.color[:, 31, 31]
    a_front = 0.5 * math.exp(-0.25 / ((0.2 * (64 / (2 * math.tan(math.radians(30)))) / 4.0) ** 2 + 0.3))
    assert np.isclose(px[1], a_front, rtol=1e-6)                   # green = alpha_front * T(=1)
    assert px[0] < px[1] and px[2] == 0.0


def test_tile_straddling_and_sh_clamp_flags():
    kw, _ = _single([[0.0, 0.0, 0.0]], [[0.25, 0.25, 0.25]], [0.9], W=64, H=64)
    f = oracle.forward(dtype=np.float64, **kw)
    # centre 31.5, radius 9 -> tile columns/rows int((31.5-9)/16)=1 .. int((31.5+9+15)/16)=3 (exclusive): 2x2 tiles
    assert f.radii[0] == 9 and f.num_rendered == 4
    g = syn.make_gaussians(64, seed=2, log_scale_mean=math.log(0.05), extent=(1, 1, 1))
    g._features_dc[:] = -3.0                                       # forces res + 0.5 < 0 -> clamped, zero gradient
    cam = syn.orbit_camera(0, 4, 64, 64, radius=4.0)
    kw = common.scene_kwargs(g, cam, 64, 64)
    f = oracle.forward(**kw)
    vis = f.radii > 0
    assert f.saved["clamped"][vis].any() and (f.saved["rgb"][vis][f.saved["clamped"][vis] == 1] == 0).all()
    gr = oracle.backward(f, np.ones((3, 64, 64), np.float32))
    dead = vis & (f.saved["clamped"].all(axis=1))
    if dead.any():
        assert (gr["shs"][dead] == 0).all()


def test_argument_validation_like_reference():
    kw, _ = _single([[0, 0, 0.0]], [[0.1] * 3], [0.5])
    bad = dict(kw); bad["shs"] = np.zeros((1, 16, 3), np.float32)
    with pytest.raises(ValueError):
        oracle.forward(**bad)
    bad = dict(kw); bad["cov3D_precomp"] = np.zeros((1, 6), np.float32)
    with pytest.raises(ValueError):
        oracle.forward(**bad)


def test_more_than_65536_tiles_sort_on_all_tile_bits():
    """The float32 oracle's instance sort is an LSD radix over 16-bit digits of tile<<32 | depth bits: 3 passes cover tile ids below
    65 536, a 4112 x 4096 image has 65 792 (round 3: the oracle lost the 17th tile bit and the full-size test caught the ORACLE).
    The float64 build sorts with qsort on (tile, depth, id): both must list the same Gaussians per pixel."""
    W, H = 4112, 4096
    g = syn.make_gaussians(300, seed=12, log_scale_mean=math.log(0.05), opacity_mean=0.0)
    cam = syn.orbit_camera(2, 9, W, H)
    kw = common.scene_kwargs(g, cam, W, H, deg=1, bg=(0.0, 0.0, 0.0))
    a = oracle.forward(count=True, **kw)
    b = oracle.forward(count=True, dtype=np.float64, **kw)
    assert a.num_rendered == b.num_rendered and a.num_rendered > 300
    assert np.array_equal(a.radii, b.radii)
    assert np.abs(a.count.astype(np.int64) - b.count).max() <= max(2, int(2e-4 * b.count.max()))   # (float32 vs float64 thresholds)
    assert np.array_equal(oracle.last_contributor_ids(a) != 0xFFFFFFFF, oracle.last_contributor_ids(b) != 0xFFFFFFFF) or \
        np.mean((oracle.last_contributor_ids(a) != 0xFFFFFFFF) != (oracle.last_contributor_ids(b) != 0xFFFFFFFF)) < 1e-5
    # (a pixel may differ by one alpha >= 1/255 decision between float32 and float64: <= 1/255 of a colour; a lost tile bit
    #  moves whole tiles: thousands of pixels by O(0.1))
    assert np.abs(a.color.astype(np.float64) - b.color).max() < 5e-3 and np.mean(np.abs(a.color - b.color) > 1e-5) < 1e-4
