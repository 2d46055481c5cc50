#!/usr/bin/env python3
"""Generate tests/golden/reference_python.npz by IMPORTING the reference's own Python
(/root/reference) in this container.  These are the only pieces of the hot path whose
arithmetic the reference owns on disk:

  utils/sh_utils.py        eval_sh, RGB2SH                         (degrees 0..3)
  utils/general_utils.py   build_rotation, build_scaling_rotation, strip_symmetric
                           (-> GaussianModel.get_covariance, scene/gaussian_model.py:29-33)
  utils/graphics_utils.py  getWorld2View2, getProjectionMatrix
  scene/cameras.py:70-85   the world_view / full_proj / camera_center construction (restated here
                           line by line because scene/cameras.py hard-codes .cuda())
  prune.py:112-128         calculate_v_imp_score  (loaded from source text; prune.py imports icecream)
  scene/gaussian_model.py:776-782  prune_gaussians' mask rule (restated: the class needs simple_knn)

The reference cannot travel to the GPU box, so the vectors are committed; rerun this script
to regenerate them:  python tests/golden/make_golden.py
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_python.npz")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sh_utils = load("utils/sh_utils.py", "ref_sh_utils")
    graphics = load("utils/graphics_utils.py", "ref_graphics_utils")
    # general_utils allocates with device="cuda": run it on CPU by redirecting the device kw
    _zeros = torch.zeros
    torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    general = load("utils/general_utils.py", "ref_general_utils")

    g = torch.Generator().manual_seed(1234)
    out = {}
    # --- SH ---
    N = 257
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g))
    sh = torch.randn(N, 3, 16, generator=g) * 0.3          # reference layout for eval_sh: [..., C, coeffs]
    out["sh_dirs"] = dirs.numpy(); out["sh_coeffs"] = sh.numpy()
    for deg in range(4):
        out[f"sh_eval_deg{deg}"] = sh_utils.eval_sh(deg, sh, dirs).numpy()
    rgb = torch.rand(64, 3, generator=g)
    out["rgb"] = rgb.numpy(); out["rgb2sh"] = sh_utils.RGB2SH(rgb).numpy()
    # --- covariance ---
    scaling = torch.exp(torch.randn(N, 3, generator=g) * 0.7 - 3.0)
    rot = torch.randn(N, 4, generator=g)
    for mod in (1.0, 0.5):
        L = general.build_scaling_rotation(mod * scaling, rot)
        cov = general.strip_symmetric(L @ L.transpose(1, 2))
        out[f"cov3d_mod{mod}"] = cov.numpy()
    out["cov_scaling"] = scaling.numpy(); out["cov_rotation"] = rot.numpy()
    out["build_rotation"] = general.build_rotation(rot).numpy()
    torch.zeros = _zeros
    # --- cameras (scene/cameras.py:64-85) ---
    cams = []
    for k in range(5):
        ang = 0.7 * k + 0.3
        c, s_ = math.cos(ang), math.sin(ang)
        R = np.array([[c, 0.1 * s_, s_], [0.05, 1.0, -0.02], [-s_, 0.0, c]])
        R, _ = np.linalg.qr(R)                           # orthonormal camera-to-world rotation
        T = np.array([0.3 * k - 0.5, 0.2, 4.0 + k])
        FoVx, FoVy = 0.9 + 0.1 * k, 0.6 + 0.05 * k
        wv = torch.tensor(graphics.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = graphics.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=FoVx, fovY=FoVy).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wv.inverse()[3, :3]
        cams.append((R, T, FoVx, FoVy, wv.numpy(), proj.numpy(), full.numpy(), center.numpy()))
    out["cam_R"] = np.stack([c[0] for c in cams]); out["cam_T"] = np.stack([c[1] for c in cams])
    out["cam_fov"] = np.array([[c[2], c[3]] for c in cams])
    out["cam_world_view"] = np.stack([c[4] for c in cams]); out["cam_proj"] = np.stack([c[5] for c in cams])
    out["cam_full_proj"] = np.stack([c[6] for c in cams]); out["cam_center"] = np.stack([c[7] for c in cams])
    out["fov2focal"] = np.array([graphics.fov2focal(0.9, 1920), graphics.focal2fov(1500.0, 1080)])
    # --- prune epilogue: execute the reference's own function text ---
    src = open(os.path.join(REF, "prune.py")).read()
    start = src.index("def calculate_v_imp_score"); end = src.index("def prune_list")
    ns = {"torch": torch}
    exec(src[start:end], ns)
    scal = torch.exp(torch.randn(5000, 3, generator=g) * 0.8 - 4.0)
    imp = torch.rand(5000, generator=g) * 100
    imp[torch.rand(5000, generator=g) < 0.2] = 0.0        # many never-hit Gaussians (ties at 0)
    gm = types.SimpleNamespace(get_scaling=scal)
    out["prune_scaling"] = scal.numpy(); out["prune_imp"] = imp.numpy()
    for v_pow in (0.1, 0.5):
        v = ns["calculate_v_imp_score"](gm, imp, v_pow)
        out[f"prune_v_list_{v_pow}"] = v.numpy()
        for pct in (0.66, 0.1):
            # scene/gaussian_model.py:776-782
            sorted_tensor, _ = torch.sort(v, dim=0)
            thr = sorted_tensor[int(pct * (sorted_tensor.shape[0] - 1))]
            out[f"prune_mask_{v_pow}_{pct}"] = (v <= thr).squeeze().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
