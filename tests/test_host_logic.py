"""CPU: host-side decisions of the Python mirrors that do not need a GPU -- which models render() may fuse, the
leaf views of backward_over_views, argument validation of the loss / prune / kNN wrappers (no CPU fallback anywhere)."""
import pytest
import torch

from lightgaussian_amd import loss_utils, parallel, prune, synthetic as syn
from lightgaussian_amd.gaussian_renderer import _has_reference_getters, _screenspace_points


def test_only_models_with_the_reference_activations_are_fused():
    g = syn.make_gaussians(50, sh_degree=2)
    assert _has_reference_getters(g)
    assert not _has_reference_getters(prune._FrozenGetters(g))          # hoisted getters: already activated tensors

    class Custom(type(g)):
        scaling_activation = staticmethod(torch.nn.functional.softplus)  # a model with its own activation keeps the literal path
    c = Custom(g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity, 2, 2)
    assert not _has_reference_getters(c)

    class NoRaw:
        get_xyz = g._xyz
    assert not _has_reference_getters(NoRaw())


def test_leaf_views_share_storage_and_are_recognised():
    g = syn.make_gaussians(20, sh_degree=1).requires_grad_(True)
    v = parallel._LeafView(g)
    assert v._xyz.data_ptr() == g._xyz.data_ptr() and v._xyz.is_leaf and v._xyz.requires_grad and v._xyz.grad_fn is None
    assert _has_reference_getters(v) and v.get_features.shape == (20, 4, 3)
    assert torch.equal(v.get_scaling, g.get_scaling.detach())


def test_screenspace_points_is_a_zero_leaf_that_collects_grad():
    g = syn.make_gaussians(7)
    sp = _screenspace_points(g)
    assert sp.is_leaf and sp.requires_grad and sp.shape == g._xyz.shape and float(sp.detach().abs().sum()) == 0.0
    (sp * 2.0).sum().backward()
    assert torch.equal(sp.grad, torch.full_like(sp, 2.0))               # gaussian_renderer/__init__.py:37-46 consumer reads .grad


def test_wrappers_refuse_cpu_tensors_and_bad_shapes():
    x = torch.rand(3, 8, 8)
    for fn in (loss_utils.l1_loss, loss_utils.ssim, loss_utils.l1_loss_only):
        with pytest.raises(RuntimeError):
            fn(x, x.clone())
    with pytest.raises(RuntimeError):
        loss_utils.l1_dssim_loss(x, x.clone(), 0.2)
    g = syn.make_gaussians(10)
    with pytest.raises(RuntimeError):
        prune.prune_epilogue(g, torch.rand(10), 0.1, 0.5)
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.rand(5, 3))
    assert float(loss_utils.l2_loss(x, x)) == 0.0                      # plain torch expression, device-agnostic


def test_shard_bounds_cover_every_view_once():
    for V in (1, 7, 200):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = prune.shard_bounds(V, world, r)
                seen += list(range(lo, hi))
            assert seen == list(range(V))
