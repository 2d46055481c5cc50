"""The PRODUCT's device-math header (lightgaussian_amd/csrc/lg_math.h), compiled for the CPU by
tests/cpu_harness, against the oracle -- bit for bit.  This is what lets us claim, without a GPU, that
(a) the canonical arithmetic of the kernels equals the oracle's, (b) the O(#binades) seqsum32 equals c
sequential float adds, (c) the exact footprint culling never changes a result.  No GPU."""
import math

import numpy as np
import pytest
import torch

import common
from common import syn
from oracle import oracle


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def test_canonical_exp_bitwise_and_accuracy():
    lib = common.harness()
    xs = np.concatenate([-np.random.RandomState(0).rand(20000) * 20, [0.0, -1e-9, -87.0, -100.0, -5.54, -1e-3]]).astype(np.float32)
    for x in xs:
        a, b = lib.h_exp(float(x)), oracle.exp_canonical(float(x))
        assert _bits(a) == _bits(b)
        if x > -80:
            assert abs(a / math.exp(float(x)) - 1) < 2.5e-7
    assert lib.h_exp(0.0) == 1.0


def test_seqsum32_equals_sequential_adds():
    lib = common.harness()
    rs = np.random.RandomState(1)
    ws = list(rs.rand(200).astype(np.float32)) + [np.float32(v) for v in
          (0.75, 0.5, 1.0, 0.1, 1.5 * 2 ** -10, 3 * 2 ** -7, (2 ** 23 + 1) * 2.0 ** -30, 2 ** -20, 0.3333, 0.99, 1e-3, 5e-8, 1e-40, 0.0,
           1 - 2 ** -24, 2 ** -24, 0.0039215689)]
    for w in ws:
        for c in list(rs.randint(0, 3000, 5)) + [0, 1, 2, 3, 4, 5, 255, 256, 100000, 2073600]:
            a = lib.h_seqsum32(float(w), int(c)); b = oracle.seqsum(float(w), int(c))
            assert _bits(a) == _bits(b), (w, c, a, b)


def test_seqsum32_random_weights_and_counts():
    lib = common.harness()
    rs = np.random.RandomState(9)
    for _ in range(400):
        w = np.float32(np.exp(rs.uniform(np.log(1e-6), np.log(1.0))))
        c = int(np.exp(rs.uniform(0, np.log(300000))))
        a = lib.h_seqsum32(float(w), c); b = oracle.seqsum(float(w), c)
        assert _bits(a) == _bits(b), (w, c, a, b)


def test_fix40_quantisation_and_score_equal_the_oracle():
    """The Q24.40 quantisation of a per-hit weight ((double) w + 4096, mantissa bits: lg_math.h) against the oracle's
    llrint(ldexp(w, 40)), and the u64 -> fp32 score conversion, bit for bit; plus the closed-form properties the design relies on."""
    lib = common.harness()
    rs = np.random.RandomState(4)
    ws = [np.float32(np.exp(rs.uniform(np.log(3.9e-7), np.log(0.99)))) for _ in range(4000)]
    ws += [np.float32(v) for v in (0.0, 0.99, 1 / 255, 1e-4 / 255, 2.0 ** -17, 2.0 ** -17 - 2.0 ** -41, 2.0 ** -18 + 2.0 ** -41, 3 * 2.0 ** -41,
                                   2.0 ** -41, 2.0 ** -22 + 2.0 ** -45, 0.5, 0.25 + 2.0 ** -26, 3.9215686e-7)]
    for w in ws:
        a, b = lib.h_fix40_quant(float(w)), oracle.fix40_quant(float(w))
        assert a == b, (w, a, b)
        if w >= 2.0 ** -17 or w == 0:
            assert a == int(float(w) * 2 ** 40)          # exactly representable: no quantisation at all
        assert abs(a - float(w) * 2 ** 40) <= 0.5
    qs = [int(v) for v in rs.randint(0, 2 ** 62, 2000, dtype=np.int64)] + [0, 1, 2 ** 24 - 1, 2 ** 24 + 1, 2 ** 40, 2 ** 63 + 2 ** 39, 2 ** 64 - 1,
                                                                         (2 ** 24 + 1) << 16, ((2 ** 24 + 1) << 16) + 1, (2 ** 25 + 3) << 20]
    for q in qs:
        a, b = lib.h_fix40_score(q), oracle.fix40_score(q)
        assert _bits(a) == _bits(b), (q, a, b)
        assert _bits(a) == _bits(np.float32(np.float32(np.uint64(q)) * np.float32(2.0 ** -40))), q


CASES = [dict(N=10000, W=256, H=256, seed=1, scale=0.004, opm=-1.0, ext=(4, 2.25, 4)),
         dict(N=3000, W=200, H=120, seed=2, scale=0.05, opm=1.0, ext=(2, 1.2, 2)),
         dict(N=800, W=128, H=96, seed=3, scale=0.3, opm=2.0, ext=(2, 1, 2)),
         dict(N=2000, W=161, H=83, seed=4, scale=0.02, opm=-3.0, ext=(3, 2, 3)),
         dict(N=1500, W=100, H=100, seed=8, scale=0.05, opm=0.5, ext=(2, 1, 2), aniso=True)]


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"N{c['N']}_{c['W']}x{c['H']}")
def test_kernel_arithmetic_and_culling_bit_exact_vs_oracle(c):
    g = syn.make_gaussians(c["N"], seed=c["seed"], log_scale_mean=math.log(c["scale"]), opacity_mean=c["opm"], extent=c["ext"],
                           log_scale_std=0.9)
    if c.get("aniso"):   # needle-like splats: the culling must fall back to the full rectangle when det cancels
        g._scaling[:, 0] += 3.0
        g._scaling[:, 1] -= 2.0
    cam = syn.orbit_camera(1, 5, c["W"], c["H"], radius=5.0)
    kw = common.scene_kwargs(g, cam, c["W"], c["H"], bg=(0.1, 0.2, 0.3))
    f = oracle.forward(count=True, **kw)
    h = common.harness_forward(kw, cull=True)
    h0 = common.harness_forward(kw, cull=False)
    assert h0["num_instances"] == f.num_rendered and h["num_instances"] <= f.num_rendered
    for out in (h, h0):
        assert np.array_equal(out["radii"], f.radii)
        assert np.array_equal(_bits(out["xy"]), _bits(f.saved["xy"]))
        assert np.array_equal(_bits(out["conic_opacity"]), _bits(f.saved["conic_opacity"]))
        assert np.array_equal(_bits(out["rgb"]), _bits(f.saved["rgb"]))
        assert np.array_equal(_bits(out["color"]), _bits(f.color))
        assert np.array_equal(out["count"], f.count)
        assert np.array_equal(_bits(out["score"]), _bits(f.score))
    vis = f.radii > 0
    tr, rr = h["tight_rect"][vis], h["ref_rect"][vis]
    assert (tr[:, 0] >= rr[:, 0]).all() and (tr[:, 1] >= rr[:, 1]).all() and (tr[:, 2] <= rr[:, 2]).all() and (tr[:, 3] <= rr[:, 3]).all()


def test_backward_geometry_stage_matches_oracle():
    """lg_backward_geom / lg_backward_sh / lg_backward_cov3d (per-Gaussian stage) vs the oracle's backward,
    fed with the oracle's own blend-stage sums (recovered from its gradients is impossible, so compare end
    to end through a colour-only loss where the blend sums are known in closed form)."""
    import ctypes as C
    lib = common.harness()
    g = syn.make_gaussians(4000, seed=21, log_scale_mean=math.log(0.03))
    cam = syn.orbit_camera(1, 6, 160, 120)
    kw = common.scene_kwargs(g, cam, 160, 120)
    f = oracle.forward(**kw)
    N, M = 4000, 16
    rs = np.random.RandomState(5)
    acc = (rs.randn(N, 9) * (f.radii[:, None] > 0)).astype(np.float32)
    # oracle side: re-run its per-Gaussian stage by calling the harness' twin with identical acc is the only
    # way to isolate the stage; the oracle has no such hook, so check the harness against float64 autograd instead
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    outs = [np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32), np.zeros((N, M, 3), np.float32), np.zeros((N, 3), np.float32),
            np.zeros((N, 4), np.float32), np.zeros((N, 6), np.float32)]
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    means3D = f32(kw["means3D"]); shs = f32(kw["shs"]); sc = f32(kw["scales"]); rot = f32(kw["rotations"])
    vm = f32(kw["viewmatrix"]); pm = f32(kw["projmatrix"]); cp = f32(kw["campos"])
    lib.h_backward_geom(N, M, 3, 160, 120, P(means3D), P(shs), P(sc), 1.0, P(rot), P(f.saved["cov3D"]), P(f.saved["clamped"]),
                        P(f.radii), P(vm), P(pm), P(cp), float(kw["tanfovx"]), float(kw["tanfovy"]), P(acc), *[P(o) for o in outs])
    # autograd reference of the same per-Gaussian map: L = sum_i acc_i . (xy_pix, A, B, C, opacity, rgb)(params)
    dd = torch.float64
    t = {k: torch.tensor(kw[k], dtype=dd, requires_grad=True) for k in ("means3D", "shs", "scales", "rotations")}
    from oracle import torch_dense as td
    vmt, pmt = torch.tensor(vm, dtype=dd), torch.tensor(pm, dtype=dd)
    ph = torch.cat([t["means3D"], torch.ones(N, 1, dtype=dd)], 1)
    pview, phom = ph @ vmt, ph @ pmt
    ndc = phom[:, :2] / (phom[:, 3:4] + 1e-7)
    ix = ((ndc[:, 0] + 1) * 160 - 1) * 0.5; iy = ((ndc[:, 1] + 1) * 120 - 1) * 0.5
    q = t["rotations"]; r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                      2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(N, 3, 3)
    L = Rm * t["scales"][:, None, :]; Sig = L @ L.transpose(1, 2)
    tz = pview[:, 2]; fx, fy = 160 / (2 * kw["tanfovx"]), 120 / (2 * kw["tanfovy"])
    limx, limy = 1.3 * kw["tanfovx"], 1.3 * kw["tanfovy"]
    txtz, tytz = pview[:, 0] / tz, pview[:, 1] / tz
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz).detach(), pview[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz).detach(), pview[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).view(N, 2, 3)
    T2 = J @ vmt[:3, :3].t(); cov = T2 @ Sig @ T2.transpose(1, 2)
    a_, b_, c_ = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a_ * c_ - b_ * b_
    A, B, Cc = c_ / det, -b_ / det, a_ / det
    d = t["means3D"] - torch.tensor(cp, dtype=dd)[None]; d = d / d.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(td.eval_sh(3, t["shs"], d) + 0.5, 0.0)
    at = torch.tensor(acc, dtype=dd); vis = torch.tensor(f.radii > 0)
    loss = (at[:, 0] * ix + at[:, 1] * iy + at[:, 2] * A + at[:, 3] * B + at[:, 4] * Cc + (at[:, 6:9] * rgb).sum(1))[vis].sum()
    loss.backward()
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    assert rel(outs[1], t["means3D"].grad.numpy()) < 5e-4
    assert rel(outs[2], t["shs"].grad.numpy()) < 1e-5
    assert rel(outs[3], t["scales"].grad.numpy()) < 5e-4
    assert rel(outs[4], t["rotations"].grad.numpy()) < 5e-4
    # NDC mean gradient = pixel gradient * (S/2)
    assert np.allclose(outs[0][:, 0], acc[:, 0] * 80.0, rtol=1e-6) and np.allclose(outs[0][:, 1], acc[:, 1] * 60.0, rtol=1e-6)
