#!/usr/bin/env python3
"""Extended seeded fuzz (not part of the test suite: ~1-2 minutes): random scene statistics incl. tiny images, single
Gaussians, depth slabs (long runs of equal sorted key bits), huge splats; hit counts / scores / radii / count-render image
bit-identical to the float oracle, training render within 1e-5, gradients within max(1e-4, 3 x fp32-oracle noise floor) of the float64 oracle.
Phase 2: getters inside the kernels against the literal getter pattern.  Phase 3 (LG_FUZZ_N3): the significance-only pass with the serial and
the parallel long-tile walk at short segment lengths, counts and scores bit-identical to the oracle.  LG_FUZZ_FIRST: first trial number."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import gpu_common  # noqa: E402
from common import syn  # noqa: E402
from oracle import oracle  # noqa: E402

if os.environ.get("LG_FUZZ_SEG"):      # e.g. 64: every list longer than 64 entries goes through the segmented backward
    from lightgaussian_amd import rasterizer as _r
    _r.set_option("segment_length", int(os.environ["LG_FUZZ_SEG"]))
if os.environ.get("LG_FUZZ_LONG"):     # serial | auto | parallel: walk of multi-segment lists in the training forward
    from lightgaussian_amd import rasterizer as _r
    _r.set_option("long_tiles", os.environ["LG_FUZZ_LONG"])
if os.environ.get("LG_FUZZ_SYNC"):     # off | validated | nowait
    from lightgaussian_amd import rasterizer as _r
    _r.set_option("sync_free", {"off": False, "validated": "validated"}[os.environ["LG_FUZZ_SYNC"]])
def make_trial(t):
    """Trial t of phase 1: (kwargs for the rasterizer as torch tensors, the same as numpy, a description, the trial's RandomState --
    positioned where the gradient image is drawn next)."""
    rs = np.random.RandomState(777 + 7919 * t)          # per-trial stream: `only` reruns exactly one trial
    N = int(rs.choice([1, 2, 7, 64, 65, 300, 2000, 9000]))
    W, H = int(rs.choice([1, 5, 16, 17, 33, 100, 257])), int(rs.choice([1, 3, 16, 31, 64, 130]))
    stored = int(rs.randint(0, 4))
    deg = int(rs.randint(0, stored + 1))                 # r4: active degree <= stored degree (scene/gaussian_model.py:125-127)
    mod = float(rs.choice([1.0, 1.0, 0.5, 2.0, np.exp(rs.uniform(np.log(0.3), np.log(3.0)))]))   # r4: scaling_modifier (gaussian_renderer/__init__.py:58)
    scale = float(np.exp(rs.uniform(np.log(0.002), np.log(0.8))))
    g = syn.make_gaussians(N, sh_degree=stored, seed=1000 + t, log_scale_mean=math.log(scale), opacity_mean=float(rs.uniform(-4, 3)),
                           extent=(float(rs.uniform(0.3, 3)), float(rs.uniform(0.3, 2)), float(rs.uniform(0.3, 3))), log_scale_std=float(rs.uniform(0.1, 1.2)))
    cam = syn.orbit_camera(int(rs.randint(0, 8)), 8, W, H, radius=float(rs.uniform(2.5, 7)))
    if rs.rand() < 0.25:   # depth slab in front of camera 0
        cam = syn.orbit_camera(0, 8, W, H, radius=5.0)
        g._xyz[:, 2] = float(rs.choice([0.0, 1e-6, 1e-4])) * torch.randn(N)
    kw = common.scene_kwargs(g, cam, W, H, deg=stored, bg=tuple(rs.rand(3).astype(np.float32)), as_torch=True)
    kw["sh_degree"] = deg
    kw["scale_modifier"] = mod
    npk = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in kw.items()}
    return kw, npk, dict(N=N, W=W, H=H, stored=stored, deg=deg, mod=mod, scale=scale), rs


if __name__ != "__main__":
    trials = 0                                           # imported (tools/gpu_fuzz_one.py): only make_trial is wanted
else:
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
only = int(sys.argv[2]) if len(sys.argv) > 2 and __name__ == "__main__" else -1
first = int(os.environ.get("LG_FUZZ_FIRST", "0"))      # trials first .. first + trials - 1: campaigns beyond the round script's 0 .. 149
bad = 0
for t in range(first, first + trials):
    if only >= 0 and t != only:
        continue
    kw, npk, meta, rs = make_trial(t)
    N, W, H, stored, deg, mod, scale = (meta[k] for k in ("N", "W", "H", "stored", "deg", "mod", "scale"))
    ref = oracle.forward(count=True, **npk)
    out = gpu_common.hip_forward_backward(kw, count=True)
    why = []
    if not np.array_equal(out["radii"], ref.radii): why.append("radii")
    if not np.array_equal(out["count"], ref.count): why.append(f"count({int((out['count'] != ref.count).sum())})")
    if not np.array_equal(out["score"].view(np.uint32), ref.score.view(np.uint32)): why.append("score")
    if not np.array_equal(out["color"].view(np.uint32), ref.color.view(np.uint32)): why.append("count-image")
    ok = not why
    gimg = rs.randn(3, H, W).astype(np.float32)
    fast = gpu_common.hip_forward_backward(kw, grad_image=gimg)
    if deg < stored and np.count_nonzero(fast["grads"]["shs"][:, (deg + 1) ** 2:]): why.append("dL_dshs beyond the active degree not zero")
    if np.abs(fast["color"] - ref.color).max() > 1e-5: why.append(f"fast-image({np.abs(fast['color'] - ref.color).max():.2e})")
    # gradients: against the float64 oracle, tolerance 1e-4 widened to 3x the error the float32 ORACLE itself has against
    # float64 on these inputs (the T/(1-alpha) replay of the published algorithm amplifies rounding), as tests/ do
    g32 = oracle.backward(ref, gimg)
    ref64 = oracle.forward(dtype=np.float64, **npk); g64 = oracle.backward(ref64, gimg)
    for name, gv in fast["grads"].items():
        if name in g64 and gv is not None and g64[name] is not None:
            r = g64[name]
            floor = gpu_common.rel_err(g32[name], r)
            e = gpu_common.rel_err(gv.reshape(r.shape), r)
            if not np.isfinite(gv).all() or e > max(1e-4, 3.0 * floor): why.append(f"grad:{name}(err {e:.2e}, fp32 floor {floor:.2e})")
    if why and only >= 0:   # diagnosis: the same gradients with the canonical arithmetic (no hardware exp / rcp)
        from lightgaussian_amd import rasterizer
        rasterizer.set_option("fast_exp", False)
        ex = gpu_common.hip_forward_backward(kw, grad_image=gimg)
        rasterizer.set_option("fast_exp", True)
        print("   canonical arithmetic:", {n: f"{gpu_common.rel_err(v.reshape(g64[n].shape), g64[n]):.2e}" for n, v in ex["grads"].items() if n in g64 and g64[n] is not None})
    if why:
        bad += 1
        print(f"MISMATCH trial {t}: N={N} {W}x{H} deg={deg}/{stored} mod={mod:.3f} scale={scale:.4f}: {', '.join(why)}")
# ---- phase 2: getters inside the kernels (render_fused, LG_FLAG_RAW_PARAMS) against the literal torch getter pattern ----
from lightgaussian_amd.gaussian_renderer import render_fused, _render_unfused  # noqa: E402
dev = torch.device("cuda:0")
bad2 = 0
if os.environ.get("LG_FUZZ_NARROW"):   # keys laid out as if only 40 bits were available (option narrow_key)
    from lightgaussian_amd import rasterizer as _r
    _r.set_option("narrow_key", True)
n2 = 0 if __name__ != "__main__" else int(os.environ.get("LG_FUZZ_N2", max(10, trials // 4)))   # LG_FUZZ_N2: size of the fused-getter phase on its own
for t in range(first, first + n2):
    rs = np.random.RandomState((991 + 104729 * t) % (2 ** 32))
    N = int(rs.choice([1, 63, 64, 65, 500, 4099, 20000]))
    W, H = int(rs.choice([16, 33, 100, 257])), int(rs.choice([16, 31, 64, 130]))
    deg = int(rs.randint(0, 4))
    act = int(rs.randint(0, deg + 1))
    mod2 = float(rs.choice([1.0, 1.0, 0.5, 2.0, 1.37]))
    mk = lambda: syn.make_gaussians(N, sh_degree=deg, seed=5000 + t, log_scale_mean=math.log(float(np.exp(rs_scale))), opacity_mean=op_mean,
                                    extent=(2, 1.2, 2)).to(dev).requires_grad_(True)
    rs_scale, op_mean = rs.uniform(np.log(0.005), np.log(0.3)), float(rs.uniform(-3, 2))
    cam = syn.orbit_camera(int(rs.randint(0, 8)), 8, W, H, radius=5.0).to(dev)
    bg = torch.tensor(rs.rand(3).astype(np.float32), device=dev)
    gimg = torch.tensor(rs.randn(3, H, W).astype(np.float32), device=dev)
    res = []
    for fn in (_render_unfused, render_fused):
        g = mk()
        g.active_sh_degree = act
        pkg = fn(cam, g, syn.PipelineParams(), bg, mod2)
        (pkg["render"] * gimg).sum().backward()
        res.append((pkg["render"].detach(), pkg["radii"], [getattr(g, n).grad for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")]))
    (ia, ra, ga), (ib, rb, gb) = res
    why = []
    # radius = ceil(3 sigma): expf in the kernel vs torch.exp can land on different sides of an integer for a Gaussian or two
    dr = (ra - rb).abs()
    if int(dr.max()) > 1 or int((dr > 0).sum()) > max(1, N // 5000): why.append(f"radii({int((dr > 0).sum())} differ, max {int(dr.max())})")
    if float((ia - ib).abs().max()) > 1e-5 * max(float(ia.abs().max()), 1e-6): why.append(f"image({float((ia - ib).abs().max()):.2e})")
    for n, x, y in zip(("xyz", "dc", "rest", "scaling", "rotation", "opacity"), ga, gb):
        if x is None or x.numel() == 0:
            continue
        e = float((x - y).abs().max()) / max(float(x.abs().max()), 1e-20)
        if y is None or not torch.isfinite(y).all() or e > 2e-4: why.append(f"grad:{n}({e:.2e})")
    if why:
        bad2 += 1
        print(f"FUSED MISMATCH trial {t}: N={N} {W}x{H} deg={act}/{deg} mod={mod2}: {', '.join(why)}")
# ---- phase 3: the significance-only pass (colours skipped, as prune_list_sharded issues it) with the serial walk and with the parallel
# ---- long-tile walk (lg_count_seg / _rewalk / _fixup, round 5) at short segment lengths: counts and scores against the oracle, bit for bit
def count_pass(kw, options):
    import ctypes as C
    from lightgaussian_amd import _lib, rasterizer
    from lightgaussian_amd.rasterizer import GaussianRasterizationSettings
    t = {k: (v.detach().to(dev).contiguous() if torch.is_tensor(v) else v) for k, v in kw.items()}
    rset = GaussianRasterizationSettings(t["H"], t["W"], t["tanfovx"], t["tanfovy"], t["bg"], t.get("scale_modifier", 1.0), t["viewmatrix"], t["projmatrix"],
                                         t["sh_degree"], t["campos"], False, False, True)
    opts = rasterizer.resolve_options(dict(options, skip_color_in_count=True, sync_free=False))
    call = rasterizer._Call(rset, t["means3D"], t.get("shs"), t.get("colors_precomp"), t["opacities"], t.get("scales"), t.get("rotations"), t.get("cov3D_precomp"),
                            exact=True, opts=opts)
    lib = _lib.load()
    _color, radii, cnt, score, _geom, binning, _img, R = rasterizer._native_forward(lib, call, rset, True)
    meta = torch.zeros(16, dtype=torch.int32, device=dev)
    _lib.check(lib.lg_debug_view_meta(C.byref(call.view), binning.data_ptr(), int(R), meta.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return cnt.cpu().numpy(), score.cpu().numpy(), meta.cpu().numpy().view(np.uint32)


bad3 = par_items = fixups = n4 = 0
n3 = 0 if __name__ != "__main__" else int(os.environ.get("LG_FUZZ_N3", trials // 2))
for t in range(first, first + n3):
    kw, npk, meta, rs = make_trial(t)
    S, wide = int(rs.choice([64, 128, 256])), bool(rs.rand() < 0.5)
    ref = oracle.forward(count=True, **npk)
    why = []
    with torch.no_grad():
        for tag, opt in (("serial", dict(count_long_tiles="serial")), ("parallel", dict(count_long_tiles="parallel", count_wide_band=wide))):
            c, sc, m = count_pass(kw, dict(opt, segment_length=S))
            if tag == "parallel":
                # (a view without instances launches no work-list workgroup: its meta words are whatever the allocator left there)
                ran = bool(0 < m[4] < (1 << 24) and m[2] == S and m[1] > 2 * S)
                par_items += int(ran); fixups += int(m[5]) if ran else 0
            if not np.array_equal(c, ref.count): why.append(f"{tag}:count({int((c != ref.count).sum())})")
            if not np.array_equal(sc.view(np.uint32), ref.score.view(np.uint32)): why.append(f"{tag}:score")
        # round 6: the per-hit weight policies (Q24.40 sums, one packed atomic per (wave, entry) into per-instance slots): counts and scores
        # against the oracle's sequential loop, bit for bit, and a second run of the same view against the first (no float atomics)
        for pol, opol in (("alpha", oracle.W_ALPHA), ("alpha_t", oracle.W_ALPHA_T)):
            rp = oracle.forward(count=True, weight_policy=opol, **npk)
            c, sc, _m = count_pass(kw, dict(weight_policy=pol, segment_length=S))
            c2, sc2, _m = count_pass(kw, dict(weight_policy=pol, segment_length=S))
            n4 += 1
            if not np.array_equal(c, rp.count): why.append(f"{pol}:count({int((c != rp.count).sum())})")
            if not np.array_equal(sc.view(np.uint32), rp.score.view(np.uint32)): why.append(f"{pol}:score({int((sc.view(np.uint32) != rp.score.view(np.uint32)).sum())})")
            if not (np.array_equal(c, c2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))): why.append(f"{pol}:run-to-run")
    if why:
        bad3 += 1
        print(f"COUNT MISMATCH trial {t}: N={meta['N']} {meta['W']}x{meta['H']} S={S} wide={wide}: {', '.join(why)}")
if __name__ == "__main__":
    print(f"fuzz: trials {first}..{first + trials - 1}, {bad} mismatches; fused-getter phase: {n2} trials, {bad2} mismatches; "
          f"significance-only phase: {n3} trials ({par_items} with multi-segment lists in the parallel walk, {fixups} exact fix-ups; {n4} per-hit-weight renders), {bad3} mismatches")
    sys.exit(1 if bad + bad2 + bad3 else 0)
