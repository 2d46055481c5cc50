"""-m gpu: what lightgaussian_amd.run.patch_reference() rebinds, run on the device.

/root/reference does not exist on the GPU box, so the reference's modules are stood in for by modules OF THE SAME NAMES that
hold the reference's literal torch formulations (each one is pinned against the reference's own file elsewhere in the suite):
    gaussian_renderer.render / count_render      getters in torch on every call, activated tensors into the rasterizer
                                                 (gaussian_renderer/__init__.py:22-124 == lightgaussian_amd _render_unfused;
                                                 call contract pinned by tests/test_dropin_reference_modules.py)
    utils.loss_utils.l1_loss / ssim              utils/loss_utils.py:18-19, 26-85 in torch ops (pinned: tests/golden/reference_loss.npz)
    prune.prune_list / calculate_v_imp_score     prune.py:112-157 (pinned: tests/test_prune_host.py, test_dropin_reference_modules.py)
    scene.gaussian_model.GaussianModel           prune_points / prune_gaussians bodies of scene/gaussian_model.py:564-600,776-782
    vectree.vq                                   the two search sites (vq.py:131-137 kmeans, :262-266 EuclideanCodebook.forward)
A "trainer" module imports those names BEFORE the patch, as prune_finetune.py:15-17,39 does; the test then runs the body of
prune_finetune.py:150-170 (render -> L1 + lambda DSSIM -> backward), the prune pass (:213-224) and the VecTree search through the
trainer's names, unpatched and patched, and compares: loss and gradients <= 1e-4, counts / scores / v_list / masks / optimizer
state bit-identical, code indices identical up to exact distance ties."""
import math
import sys
import types

import pytest
import torch
from torch import nn

from common import syn
from lightgaussian_amd import run as lg_run

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
_NAMES = ("gaussian_renderer", "utils", "utils.loss_utils", "prune", "scene", "scene.gaussian_model", "vectree", "vectree.vq", "fake_trainer")

_VQ_SRC = '''
import torch
import torch.nn.functional as F

def gumbel_sample(t, temperature = 1., dim = -1):
    if temperature == 0:
        return t.argmax(dim = dim)
    raise NotImplementedError

def kmeans_assign(samples, means):          # vectree/vq.py:131-137
    dists = -torch.cdist(samples, means, p = 2)
    buckets = torch.argmax(dists, dim = -1)
    return buckets

class EuclideanCodebook(torch.nn.Module):
    def __init__(self, embed):
        super().__init__()
        self.register_buffer("embed", embed)
        self.sample_codebook_temp = 0
    def forward(self, x):                   # vectree/vq.py:258-269, the search and the lookup
        flatten = x.float()
        dist = -torch.cdist(flatten, self.embed, p = 2)
        embed_ind = gumbel_sample(dist, dim = -1, temperature = self.sample_codebook_temp)
        quantize = torch.stack([self.embed[i][embed_ind[i]] for i in range(self.embed.shape[0])])
        return quantize, embed_ind
'''


class _Model(syn.SyntheticGaussians):
    """SyntheticGaussians (raw parameters + the reference's getters) with the optimizer / bookkeeping surface and the literal
    prune_points / prune_gaussians bodies of scene/gaussian_model.py:564-600, 776-782."""

    def setup(self):
        for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            setattr(self, n, nn.Parameter(getattr(self, n).detach().clone().requires_grad_(True)))
        groups = [{"params": [self._xyz], "lr": 1e-3, "name": "xyz"}, {"params": [self._features_dc], "lr": 1e-3, "name": "f_dc"},
                  {"params": [self._features_rest], "lr": 1e-4, "name": "f_rest"}, {"params": [self._opacity], "lr": 1e-2, "name": "opacity"},
                  {"params": [self._scaling], "lr": 1e-3, "name": "scaling"}, {"params": [self._rotation], "lr": 1e-3, "name": "rotation"}]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        g = torch.Generator().manual_seed(5)
        for grp in groups:
            grp["params"][0].grad = (torch.randn(grp["params"][0].shape, generator=g) * 1e-3).to(DEV)
        self.optimizer.step()
        for grp in groups:
            grp["params"][0].grad = None
        N = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.rand(N, 1, generator=g).to(DEV)
        self.denom = torch.rand(N, 1, generator=g).to(DEV)
        self.max_radii2D = torch.rand(N, generator=g).to(DEV)
        return self

    def _prune_optimizer(self, mask):
        optimizable_tensors = {}
        for group in self.optimizer.param_groups:
            stored_state = self.optimizer.state.get(group["params"][0], None)
            if stored_state is not None:
                stored_state["exp_avg"] = stored_state["exp_avg"][mask]
                stored_state["exp_avg_sq"] = stored_state["exp_avg_sq"][mask]
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter((group["params"][0][mask].requires_grad_(True)))
                self.optimizer.state[group["params"][0]] = stored_state
                optimizable_tensors[group["name"]] = group["params"][0]
            else:
                group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
                optimizable_tensors[group["name"]] = group["params"][0]
        return optimizable_tensors

    def prune_points(self, mask):
        valid_points_mask = ~mask
        optimizable_tensors = self._prune_optimizer(valid_points_mask)
        self._xyz = optimizable_tensors["xyz"]
        self._features_dc = optimizable_tensors["f_dc"]
        self._features_rest = optimizable_tensors["f_rest"]
        self._opacity = optimizable_tensors["opacity"]
        self._scaling = optimizable_tensors["scaling"]
        self._rotation = optimizable_tensors["rotation"]
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid_points_mask]
        self.denom = self.denom[valid_points_mask]
        self.max_radii2D = self.max_radii2D[valid_points_mask]

    def prune_gaussians(self, percent, import_score):
        sorted_tensor, _ = torch.sort(import_score, dim=0)
        index_nth_percentile = int(percent * (sorted_tensor.shape[0] - 1))
        value_nth_percentile = sorted_tensor[index_nth_percentile]
        prune_mask = (import_score <= value_nth_percentile).squeeze()
        self.prune_points(prune_mask)


def _model(N, seed):
    g = syn.make_gaussians(N, seed=seed, log_scale_mean=math.log(0.02)).to(DEV)
    return _Model(g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity, 3, 3).setup()


@pytest.fixture()
def standins():
    from lightgaussian_amd import gaussian_renderer as lg_gr, loss_utils as lg_loss, prune as lg_prune
    saved = {n: sys.modules.get(n) for n in _NAMES}
    gr = types.ModuleType("gaussian_renderer")

    def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
        return lg_gr._render_unfused(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)

    def count_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
        return lg_gr.count_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)
    gr.render, gr.count_render = render, count_render
    utils = types.ModuleType("utils"); utils.__path__ = []
    lu = types.ModuleType("utils.loss_utils")
    lu.l1_loss = lambda network_output, gt: torch.abs((network_output - gt)).mean()                           # utils/loss_utils.py:18-19
    lu.ssim = lambda img1, img2, window_size=11, size_average=True: lg_loss._ssim_general(img1, img2, window_size, size_average)
    pr = types.ModuleType("prune"); pr.__file__ = "/standin/prune.py"
    pr.prune_list = lambda gaussians, scene, pipe, background: lg_prune.prune_list(gaussians, scene, pipe, background, count_fn=gr.count_render)
    pr.calculate_v_imp_score = lg_prune.calculate_v_imp_score
    scene = types.ModuleType("scene"); scene.__path__ = []
    gm = types.ModuleType("scene.gaussian_model"); gm.GaussianModel = _Model
    vt = types.ModuleType("vectree"); vt.__path__ = []
    vq = types.ModuleType("vectree.vq")
    exec(compile(_VQ_SRC, "/standin/vectree/vq.py", "exec"), vq.__dict__)
    for name, mod in (("gaussian_renderer", gr), ("utils", utils), ("utils.loss_utils", lu), ("prune", pr), ("scene", scene),
                      ("scene.gaussian_model", gm), ("vectree", vt), ("vectree.vq", vq)):
        sys.modules[name] = mod
    trainer = types.ModuleType("fake_trainer")                      # prune_finetune.py:15-17,39: names bound at import time
    trainer.render, trainer.count_render, trainer.l1_loss, trainer.ssim = gr.render, gr.count_render, lu.l1_loss, lu.ssim
    trainer.prune_list, trainer.calculate_v_imp_score = pr.prune_list, pr.calculate_v_imp_score
    sys.modules["fake_trainer"] = trainer
    orig_pp, orig_pg = _Model.prune_points, _Model.prune_gaussians
    yield trainer, vq
    lg_run.unpatch_reference()
    assert _Model.prune_points is orig_pp and _Model.prune_gaussians is orig_pg
    for n, m in saved.items():
        if m is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = m


def _step(trainer, model, cam, gt, pipe, bg, lambda_dssim=0.2):
    """prune_finetune.py:150-170"""
    for p in (model._xyz, model._features_dc, model._features_rest, model._opacity, model._scaling, model._rotation):
        p.grad = None
    render_pkg = trainer.render(cam, model, pipe, bg)
    image, viewspace_point_tensor, visibility_filter, radii = (render_pkg["render"], render_pkg["viewspace_points"],
                                                               render_pkg["visibility_filter"], render_pkg["radii"])
    Ll1 = trainer.l1_loss(image, gt)
    loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - trainer.ssim(image, gt))
    loss.backward()
    grads = {n: getattr(model, n).grad.detach().clone() for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}
    grads["viewspace"] = viewspace_point_tensor.grad.detach().clone()
    return float(loss), float(Ll1), image.detach().clone(), radii.clone(), visibility_filter.clone(), grads


def test_training_step_through_the_patched_names_equals_the_literal_path(standins):
    trainer, _vq = standins
    W, H = 480, 320
    model = _model(30000, 4)
    cams = [syn.orbit_camera(k, 5, W, H).to(DEV) for k in range(3)]
    pipe, bg = syn.PipelineParams(), torch.zeros(3, device=DEV)
    with torch.no_grad():
        pert = syn.make_gaussians(30000, seed=4, log_scale_mean=math.log(0.021)).to(DEV)
        gts = [trainer.render(c, pert, pipe, bg)["render"].clone() for c in cams]
    literal = [_step(trainer, model, c, g, pipe, bg) for c, g in zip(cams, gts)]
    report = lg_run.patch_reference()
    assert report["gaussian_renderer.render"]["also_rebound_in"] >= 1           # the trainer's own binding followed
    from lightgaussian_amd import gaussian_renderer as lg_gr, loss_utils as lg_loss
    assert trainer.render is lg_gr.render and trainer.ssim is lg_loss.ssim and trainer.l1_loss is lg_loss.l1_loss
    patched = [_step(trainer, model, c, g, pipe, bg) for c, g in zip(cams, gts)]
    for a, b in zip(literal, patched):
        assert abs(a[0] - b[0]) <= 1e-4 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-4 * abs(a[1])
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])     # the fused forward is bit-identical
        for n in a[5]:
            scale = float(a[5][n].abs().max())
            assert float((a[5][n] - b[5][n]).abs().max()) <= 1e-4 * scale, (n, float((a[5][n] - b[5][n]).abs().max()) / scale)
    lg_run.unpatch_reference()
    again = _step(trainer, model, cams[0], gts[0], pipe, bg)
    assert again[0] == literal[0][0]                                             # unpatched again: the literal path, bit for bit


def test_prune_pass_and_model_surgery_through_the_patched_names(standins):
    trainer, _vq = standins
    W, H = 320, 240
    cams = [syn.orbit_camera(k, 7, W, H).to(DEV) for k in range(7)]
    pipe, bg = syn.PipelineParams(), torch.zeros(3, device=DEV)

    def prune_block(model):                                                       # prune_finetune.py:213-224 ("v_important_score")
        with torch.no_grad():
            gaussian_list, imp_list = trainer.prune_list(model, cams, pipe, bg)
            v_list = trainer.calculate_v_imp_score(model, imp_list, 0.1)
            model.prune_gaussians(0.6, v_list)
        return gaussian_list.clone(), imp_list.clone(), v_list.clone()

    a, b = _model(20000, 6), _model(20000, 6)
    ca, ia, va = prune_block(a)
    lg_run.patch_reference()
    assert trainer.prune_list is lg_run._prune_list and _Model.prune_gaussians is lg_run._prune_gaussians
    cb, ib, vb = prune_block(b)
    assert torch.equal(ca.to(cb.dtype), cb) and torch.equal(ia, ib) and torch.equal(va, vb)
    assert a._xyz.shape == b._xyz.shape and 0.3 < a._xyz.shape[0] / 20000 < 0.5
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        sa, sb = a.optimizer.state[ga["params"][0]], b.optimizer.state[gb["params"][0]]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    # integer scores (prune_type "count", prune_finetune.py:229-232) keep the reference's own sort formulation
    c, d = _model(5000, 7), _model(5000, 7)
    counts = torch.randint(0, 50, (5000,), generator=torch.Generator().manual_seed(2), dtype=torch.int32).to(DEV)
    lg_run.unpatch_reference()
    c.prune_gaussians(0.5, counts)
    lg_run.patch_reference()
    d.prune_gaussians(0.5, counts)
    assert torch.equal(c._xyz, d._xyz)


def test_vectree_search_sites_through_the_patched_module(standins):
    _trainer, vq = standins
    g = torch.Generator().manual_seed(3)
    for d, K, n in ((27, 8192, 20000), (48, 8192, 9000), (12, 100, 333)):
        embed = torch.randn(1, K, d, generator=g).to(DEV)
        x = (embed[0][torch.randint(0, K, (n,), generator=g).to(DEV)] + 0.05 * torch.randn(n, d, generator=g).to(DEV)).unsqueeze(0)
        cb = vq.EuclideanCodebook(embed)
        q0, i0 = cb(x)
        k0 = vq.kmeans_assign(x, embed)
        lg_run.patch_reference()
        assert type(vq.torch).__name__ == "_TorchProxy"
        q1, i1 = cb(x)
        k1 = vq.kmeans_assign(x, embed)
        lg_run.unpatch_reference()
        assert vq.torch is torch
        # identical up to exact ties of |x - c| (the fused search evaluates |c|^2 - 2 x.c; cdist rounds differently): every
        # disagreeing row must be a tie within float rounding of the two distances
        for a, b in ((i0, i1), (k0, k1)):
            bad = (a != b).nonzero()
            assert bad.shape[0] <= 2, bad.shape[0]
            for _h, r in bad.tolist():
                da = float((x[0, r] - embed[0, a[0, r]]).norm()); db = float((x[0, r] - embed[0, b[0, r]]).norm())
                assert abs(da - db) <= 1e-5 * max(da, 1e-6)
        same = (i0 == i1)[0]
        assert torch.equal(q0[0][same], q1[0][same])


def test_fused_adam_switch_gives_the_same_update_on_the_gpu():
    """run.py --fused-adam on CUDA parameters: fused=True is set on the optimizer the reference's constructor call builds, the
    moments live under the keys the reference's prune / densify surgery indexes, and three steps equal torch's default Adam to
    float rounding."""
    import torch
    from lightgaussian_amd import run as lg_run
    torch.manual_seed(0)
    base = [torch.randn(1000, 3, device="cuda:0"), torch.randn(1000, 15, 3, device="cuda:0"), torch.randn(1000, 1, device="cuda:0")]
    grads = [[torch.randn_like(b) for b in base] for _ in range(3)]

    def run(fused):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        groups = [{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(ps, (1.6e-4, 1.25e-4, 0.05), ("xyz", "f_rest", "opacity"))]
        if fused:
            lg_run.fused_adam(True)
        try:
            opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        finally:
            lg_run.fused_adam(False)
        assert bool(opt.defaults.get("fused")) == fused
        for g3 in grads:
            for p, g in zip(ps, g3):
                p.grad = g.clone()
            opt.step()
            opt.zero_grad(set_to_none=True)
        st = opt.state[ps[0]]
        assert "exp_avg" in st and "exp_avg_sq" in st
        return [p.detach() for p in ps]

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-7)
