"""Shared helpers for the test-suite (test infrastructure)."""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lightgaussian_amd import synthetic as syn  # noqa: E402


def scene_kwargs(g, cam, W, H, *, deg=3, precolor=None, precov=False, bg=(0.0, 0.0, 0.0), as_torch=False):
    """Rasterizer kwargs (numpy) for a SyntheticGaussians + MiniCam."""
    import torch
    M = (deg + 1) ** 2
    kw = dict(means3D=g.get_xyz, opacities=g.get_opacity, W=W, H=H, tanfovx=math.tan(cam.FoVx * 0.5),
              tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.tensor(bg, dtype=torch.float32),
              viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
              sh_degree=deg)
    if precolor is not None:
        kw["colors_precomp"] = precolor
    else:
        kw["shs"] = g.get_features[:, :M].contiguous()
    if precov:
        kw["cov3D_precomp"] = g.get_covariance()
    else:
        kw["scales"] = g.get_scaling
        kw["rotations"] = g.get_rotation
    if as_torch:
        return kw
    return {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in kw.items()}


_HARNESS = None


def harness():
    """g++ build of tests/cpu_harness (the product's lg_math.h compiled for the CPU)."""
    global _HARNESS
    if _HARNESS is not None:
        return _HARNESS
    d = os.path.join(ROOT, "tests", "cpu_harness")
    so = os.path.join(d, "liblg_math_harness.so")
    srcs = [os.path.join(d, "lg_math_harness.cpp"), os.path.join(ROOT, "lightgaussian_amd", "csrc", "lg_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "-mfma", "-mavx2", srcs[0], "-o", so])
    lib = C.CDLL(so)
    lib.h_exp.restype = C.c_float; lib.h_exp.argtypes = [C.c_float]
    lib.h_seqsum32.restype = C.c_float; lib.h_seqsum32.argtypes = [C.c_float, C.c_uint32]
    lib.h_fix40_quant.restype = C.c_uint64; lib.h_fix40_quant.argtypes = [C.c_float]
    lib.h_fix40_score.restype = C.c_float; lib.h_fix40_score.argtypes = [C.c_uint64]
    P = C.c_void_p
    lib.h_forward.restype = C.c_int
    lib.h_forward.argtypes = [C.c_int] * 5 + [P] * 6 + [C.c_float] + [P] * 5 + [C.c_float, C.c_float, C.c_int] + [P] * 10
    lib.h_backward_geom.restype = None
    lib.h_backward_geom.argtypes = [C.c_int] * 5 + [P, P, P, C.c_float, P, P, P, P, P, P, P, C.c_float, C.c_float, P] + [P] * 6
    _HARNESS = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def harness_forward(kw, cull=True, count=True):
    """Run the CPU emulation of the kernels' traversal; kw = scene_kwargs(...) numpy dict."""
    lib = harness()
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
    means3D = f32(kw["means3D"]); N = means3D.shape[0]
    shs = f32(kw.get("shs")); colors = f32(kw.get("colors_precomp"))
    M = 0 if shs is None else shs.shape[1]
    W, H = kw["W"], kw["H"]
    out = dict(color=np.zeros((3, H, W), np.float32), radii=np.zeros(N, np.int32), count=np.zeros(N, np.int32),
               score=np.zeros(N, np.float32), xy=np.zeros((N, 2), np.float32), conic_opacity=np.zeros((N, 4), np.float32),
               rgb=np.zeros((N, 3), np.float32), ref_rect=np.zeros((N, 4), np.int32), tight_rect=np.zeros((N, 4), np.int32))
    ninst = C.c_longlong(0)
    op = f32(kw["opacities"]).reshape(-1)
    sc = f32(kw.get("scales")); rot = f32(kw.get("rotations")); cov = f32(kw.get("cov3D_precomp"))
    bg = f32(kw["bg"]); vm = f32(kw["viewmatrix"]); pm = f32(kw["projmatrix"]); cp = f32(kw["campos"])
    lib.h_forward(N, M, int(kw["sh_degree"]), W, H, _p(bg), _p(means3D), _p(shs), _p(colors), _p(op), _p(sc),
                  float(kw.get("scale_modifier", 1.0)), _p(rot), _p(cov), _p(vm), _p(pm), _p(cp), float(kw["tanfovx"]),
                  float(kw["tanfovy"]), int(cull), _p(out["color"]), _p(out["radii"]),
                  _p(out["count"]) if count else None, _p(out["score"]) if count else None, _p(out["xy"]),
                  _p(out["conic_opacity"]), _p(out["rgb"]), _p(out["ref_rect"]), _p(out["tight_rect"]), C.byref(ninst))
    out["num_instances"] = ninst.value
    return out
