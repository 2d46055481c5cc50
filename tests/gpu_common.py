"""GPU-side helpers: run the HIP rasterizer through the drop-in API and compare with the oracle."""
import math

import numpy as np
import torch

import common
from lightgaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from oracle import oracle


def hip_forward_backward(kw_t, *, count=False, grad_image=None, debug=False):
    """kw_t: scene_kwargs(..., as_torch=True) on CPU.  Returns dict of numpy outputs (+grads)."""
    dev = torch.device("cuda:0")
    t = {k: (v.detach().to(dev).clone() if torch.is_tensor(v) else v) for k, v in kw_t.items()}
    names = ["means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]
    for n in names:
        if n in t and grad_image is not None:
            t[n].requires_grad_(True)
    N = t["means3D"].shape[0]
    means2D = torch.zeros((N, 3), device=dev, requires_grad=grad_image is not None)
    rs = GaussianRasterizationSettings(
        image_height=t["H"], image_width=t["W"], tanfovx=t["tanfovx"], tanfovy=t["tanfovy"], bg=t["bg"],
        scale_modifier=t.get("scale_modifier", 1.0), viewmatrix=t["viewmatrix"], projmatrix=t["projmatrix"],
        sh_degree=t["sh_degree"], campos=t["campos"], prefiltered=False, debug=debug, f_count=count)
    rast = GaussianRasterizer(raster_settings=rs)
    out = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"),
               colors_precomp=t.get("colors_precomp"), scales=t.get("scales"), rotations=t.get("rotations"),
               cov3D_precomp=t.get("cov3D_precomp"))
    res = {}
    if count:
        cnt, score, color, radii = out
        res["count"] = cnt.cpu().numpy(); res["score"] = score.cpu().numpy()
    else:
        color, radii = out
    res["color"] = color.detach().cpu().numpy(); res["radii"] = radii.cpu().numpy()
    if grad_image is not None:
        (color * torch.as_tensor(grad_image, device=dev)).sum().backward()
        res["grads"] = {"means2D": means2D.grad.cpu().numpy()}
        for n in names:
            if n in t:
                res["grads"][n] = t[n].grad.cpu().numpy()
    torch.cuda.synchronize()
    return res


def rel_err(a, b):
    """max |a-b| relative to max |b| (tensor-level relative error used for the 1e-4 contract)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def elem_excess(a, b, rtol=1e-4, atol_frac=2e-5):
    """Element-wise check |a - b| <= rtol |b| + atol with atol = atol_frac * max|b| (an absolute floor for entries that are
    small only through cancellation).  Returns (worst ratio |a-b| / bound, flat index of the worst element): <= 1 passes."""
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    bound = rtol * np.abs(b) + atol_frac * (np.abs(b).max() + 1e-300)
    ratio = np.abs(a - b) / bound
    i = int(ratio.argmax()) if ratio.size else 0
    return (float(ratio[i]) if ratio.size else 0.0), i
