"""-m gpu: compaction after a prune (lg_compact_plan / lg_compact_rows, prune.prune_points) against torch boolean indexing
and against the reference's own optimizer surgery, restated literally from scene/gaussian_model.py:564-600 -- bit-equal."""
import numpy as np
import pytest
import torch
from torch import nn

from lightgaussian_amd import prune

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N,frac,seed", [(1, 1.0, 0), (1, 0.0, 1), (1023, 0.5, 2), (1024, 0.34, 3), (1025, 0.9, 4), (300_000, 0.34, 5),
                                         (3_000_000, 0.34, 6), (5000, 0.0, 7), (5000, 1.0, 8)])
def test_compact_tensors_equals_boolean_indexing(N, frac, seed):
    g = torch.Generator().manual_seed(seed)
    keep = (torch.rand(N, generator=g) < frac).to(DEV)
    shapes = [(N, 3), (N, 1, 3), (N, 15, 3), (N, 1), (N, 4), (N,)]
    ts = [torch.randn(*s, generator=g).to(DEV) for s in shapes] + [torch.randint(0, 1000, (N,), generator=g, dtype=torch.int32).to(DEV),
                                                                   torch.randint(0, 2, (N, 3), generator=g, dtype=torch.uint8).to(DEV)]
    outs = prune.compact_tensors(ts, keep)
    for t, o in zip(ts, outs):
        ref = t[keep]
        assert o.shape == ref.shape and o.dtype == ref.dtype
        assert torch.equal(o, ref)


class _Model:
    """The attribute surface GaussianModel.prune_points touches (scene/gaussian_model.py:45-60, 520-535, 584-600)."""

    def __init__(self, N, seed, steps=2):
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: nn.Parameter(torch.randn(*s, generator=g).to(DEV).requires_grad_(True))  # noqa: E731
        self._xyz, self._features_dc, self._features_rest = mk(N, 3), mk(N, 1, 3), mk(N, 15, 3)
        self._opacity, self._scaling, self._rotation = mk(N, 1), mk(N, 3), mk(N, 4)
        groups = [{"params": [self._xyz], "lr": 1e-3, "name": "xyz"}, {"params": [self._features_dc], "lr": 1e-3, "name": "f_dc"},
                  {"params": [self._features_rest], "lr": 1e-4, "name": "f_rest"}, {"params": [self._opacity], "lr": 1e-2, "name": "opacity"},
                  {"params": [self._scaling], "lr": 1e-3, "name": "scaling"}, {"params": [self._rotation], "lr": 1e-3, "name": "rotation"}]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        for _ in range(steps):                                   # populate exp_avg / exp_avg_sq
            for grp in groups:
                grp["params"][0].grad = torch.randn(grp["params"][0].shape, generator=g).to(DEV)
            self.optimizer.step()
        self.xyz_gradient_accum = torch.rand(N, 1, generator=g).to(DEV)
        self.denom = torch.rand(N, 1, generator=g).to(DEV)
        self.max_radii2D = torch.rand(N, generator=g).to(DEV)

    # scene/gaussian_model.py:564-600, literally
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

    def reference_prune_points(self, mask):
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


def _state(m):
    out = {n: getattr(m, n).detach().clone() for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation",
                                                      "xyz_gradient_accum", "denom", "max_radii2D")}
    for grp in m.optimizer.param_groups:
        st = m.optimizer.state[grp["params"][0]]
        out[grp["name"] + ".exp_avg"] = st["exp_avg"].clone()
        out[grp["name"] + ".exp_avg_sq"] = st["exp_avg_sq"].clone()
        out[grp["name"] + ".step"] = torch.as_tensor(st["step"]).clone()
        assert grp["params"][0] is getattr(m, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                                               "scaling": "_scaling", "rotation": "_rotation"}[grp["name"]])
    return out


@pytest.mark.parametrize("N,with_state", [(20_000, True), (20_000, False), (257, True)])
def test_prune_points_leaves_the_state_the_reference_leaves(N, with_state):
    a, b = _Model(N, 3, steps=2 if with_state else 0), _Model(N, 3, steps=2 if with_state else 0)
    mask = (torch.rand(N, generator=torch.Generator().manual_seed(9)) < 0.66).to(DEV)
    a.reference_prune_points(mask)
    prune.prune_points(b, mask)
    if with_state:
        sa, sb = _state(a), _state(b)
        assert sa.keys() == sb.keys()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
        # the optimizer keeps working on the compacted parameters
        for grp in b.optimizer.param_groups:
            grp["params"][0].grad = torch.ones_like(grp["params"][0])
        b.optimizer.step()
    else:
        for n in ("_xyz", "_features_rest", "max_radii2D"):
            assert torch.equal(getattr(a, n), getattr(b, n)), n
    assert b._xyz.shape[0] == int((~mask).sum())


def test_prune_gaussians_threshold_and_ties():
    m = _Model(5000, 4)
    score = torch.rand(5000, generator=torch.Generator().manual_seed(1)).to(DEV)
    score[::3] = 0.0
    ref_mask = prune.prune_mask(0.5, score)
    mask = prune.prune_gaussians(m, 0.5, score)
    assert torch.equal(mask, ref_mask)
    assert m._xyz.shape[0] == int((~ref_mask).sum())
