"""Host-side mirror of the reference's render boundary (gaussian_renderer/__init__.py).

render()        <- gaussian_renderer/__init__.py:22-124
count_render()  <- gaussian_renderer/__init__.py:127-229
Same names, argument meaning, result-dict keys and error behaviour; the only change is the import
target of the rasterizer (our HIP library instead of the CUDA submodule) and that tensors are
created on the Gaussians' own device instead of the hard-coded "cuda" (identical on one GPU,
required for one-process-per-GPU sharding).
"""
import math
import threading

import torch

from . import rasterizer as _rasterizer
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_raw
from .sh_utils import eval_sh


def _settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, f_count):
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=pipe.debug,
        f_count=f_count,
    )


def _inputs(viewpoint_camera, pc, pipe, scaling_modifier, override_color):
    means3D = pc.get_xyz
    opacity = pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    return means3D, opacity, scales, rotations, cov3D_precomp, shs, colors_precomp


def _screenspace_points(pc):
    """Zero tensor whose .grad receives the 2D (NDC) mean gradients, gaussian_renderer/__init__.py:37-46.  The reference
    builds it as zeros(..., requires_grad=True) + 0 followed by retain_grad(); a plain leaf gets its .grad the same way
    and saves the add, its autograd node and the clone retain_grad() makes at the end of every backward (~25 us per step
    at 3M Gaussians)."""
    xyz = pc.get_xyz
    if not xyz.is_cuda or torch.is_inference_mode_enabled():      # (inference tensors cannot share a normal buffer's leaf machinery)
        return torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device)
    # Nothing ever writes this tensor (it exists for its .grad), so every view's leaf can share ONE zero buffer per device and
    # shape: detach() gives a fresh leaf -- its own .grad, its own identity in the render package -- over the same storage, and
    # the 36 MB fill per render (3M Gaussians: ~8 us) happens once.
    key = (xyz.device.index, tuple(xyz.shape), xyz.dtype)
    with _ZERO_LOCK:
        buf = _ZERO_POINTS.pop(key, None)
        if buf is None:
            buf = torch.zeros_like(xyz, requires_grad=False)
        _ZERO_POINTS[key] = buf                       # most recently used last
        # a few shapes per process (distill_train.py renders a teacher and a student of different N in every iteration; a prune
        # changes N): least recently used out, under the lock (backward_over_views(host_threads=True) renders from several threads)
        while len(_ZERO_POINTS) > _ZERO_KEEP:
            _ZERO_POINTS.pop(next(iter(_ZERO_POINTS)))
    return buf.detach().requires_grad_(True)


_ZERO_POINTS = {}
_ZERO_LOCK = threading.Lock()
_ZERO_KEEP = 4


_RAW_FIELDS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _has_reference_getters(pc):
    """True when `pc` is a GaussianModel whose getters are exactly the reference's (scene/gaussian_model.py:36-47 and
    :98-118: get_scaling = torch.exp(_scaling), get_opacity = torch.sigmoid(_opacity), get_rotation =
    F.normalize(_rotation), get_features = cat(_features_dc, _features_rest)) -- identified by the activation attributes
    setup_functions() installs.  Anything else (frozen getters, custom activations, foreign models) is not touched."""
    if not all(hasattr(pc, n) for n in _RAW_FIELDS):
        return False
    return (getattr(pc, "scaling_activation", None) is torch.exp and getattr(pc, "opacity_activation", None) is torch.sigmoid
            and getattr(pc, "rotation_activation", None) is torch.nn.functional.normalize
            and pc._features_rest.dim() == 3 and pc._features_dc.dim() == 3 and pc._features_dc.shape[1] == 1)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, *, options=None):
    """Render the scene.  Background tensor (bg_color) must be on the GPU.
    options (keyword-only extension): per-call overrides of the rasterizer knobs (rasterizer.set_option), e.g.
    {"sync_free": True}; the reference's six positional parameters are unchanged.

    With option fuse_getters (default True) and a GaussianModel carrying the reference's own activations, the getters
    are evaluated INSIDE the kernels from the raw parameters (render_fused: no torch.cat of the SH tensors, no
    activation kernels; same values to ~1e-7, gradients land on the raw parameters exactly as autograd would route
    them).  set_option("fuse_getters", False) restores the reference's literal call pattern."""
    if (_rasterizer.resolve_options(options)["fuse_getters"] and override_color is None and not pipe.convert_SHs_python
            and not pipe.compute_cov3D_python and _has_reference_getters(pc)):
        return render_fused(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, options=options)
    return _render_unfused(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, options)


def _render_unfused(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, options=None):
    """gaussian_renderer/__init__.py:22-124 literally: getters in torch, activated tensors into the rasterizer."""
    screenspace_points = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, False), options=options)
    means3D, opacity, scales, rotations, cov3D_precomp, shs, colors_precomp = _inputs(
        viewpoint_camera, pc, pipe, scaling_modifier, override_color)
    rendered_image, radii = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}


def count_render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, *, options=None):
    """render() + per-Gaussian hit count and Global Significance score (f_count=True).  options: as for render(), e.g.
    {"skip_color_in_count": True} for passes that only consume the counts / scores."""
    screenspace_points = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, True), options=options)
    means3D, opacity, scales, rotations, cov3D_precomp, shs, colors_precomp = _inputs(
        viewpoint_camera, pc, pipe, scaling_modifier, override_color)
    gaussians_count, important_score, rendered_image, radii = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "gaussians_count": gaussians_count, "important_score": important_score}


def render_fused(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, *, options=None):
    """render() with the getters fused into the kernels (opt-in extension -- the reference's
    render() evaluates exp/sigmoid/normalize and cat(_features_dc, _features_rest) in torch on every call, which at
    3M Gaussians costs as much HBM traffic as the whole rasterizer).  Reads GaussianModel's raw tensors
    (_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation: scene/gaussian_model.py:45-60) directly; same
    result dict, gradients land on the raw parameters.  Falls back to render() for the Python-side alternates."""
    if override_color is not None or pipe.convert_SHs_python or pipe.compute_cov3D_python:
        return _render_unfused(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, options)
    screenspace_points = _screenspace_points(pc)
    rs = _settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, False)
    rendered_image, radii, visible = rasterize_gaussians_raw(pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity,
                                                             pc._scaling, pc._rotation, rs, options)
    # visibility_filter = radii > 0 (gaussian_renderer/__init__.py:121), as K1 left it in the forward's geom buffer: no compare kernel
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": visible,
            "radii": radii}
