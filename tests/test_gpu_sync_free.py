"""-m gpu: the capacity-bounded forward (lg_forward_bounded: no read-back of the instance count) against the exact forward.
Same kernels, same total order of the tile lists => counts, scores, radii, image and gradients must be BIT-IDENTICAL; a view
that does not fit its capacity (or exceeds the depth bound) must be reported on the device and contribute zeros."""
import math

import numpy as np
import pytest
import torch

import common
from common import syn
from lightgaussian_amd import rasterizer, prune, parallel
from lightgaussian_amd.gaussian_renderer import render, count_render

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_options():
    rasterizer.set_option("sync_free", False)      # the tests below choose the mode explicitly
    rasterizer._CAPACITY.clear()
    yield
    rasterizer.set_option("sync_free", "validated")
    rasterizer.set_option("max_depth", 100.0)
    rasterizer.pending_status()
    rasterizer._CAPACITY.clear()


def _scene(N=20000, W=320, H=240, scale=0.03):
    g = syn.make_gaussians(N, seed=5, log_scale_mean=math.log(scale)).to(DEV)
    cams = [syn.orbit_camera(k, 6, W, H).to(DEV) for k in range(6)]
    return g, cams, syn.PipelineParams(), torch.tensor([0.1, 0.2, 0.3], device=DEV)


def test_bounded_forward_is_bit_identical_to_the_exact_forward():
    g, cams, pipe, bg = _scene()
    with torch.no_grad():
        exact = [count_render(c, g, pipe, bg) for c in cams]
        rasterizer.set_option("sync_free", True)
        first = count_render(cams[0], g, pipe, bg)                 # learns the capacity on the exact path
        assert rasterizer.pending_status() == [False]
        free = [count_render(c, g, pipe, bg) for c in cams]        # all bounded now
        flags = rasterizer.pending_status()
    assert flags == [False] * len(cams)
    for a, b in zip(exact, free):
        for k in ("gaussians_count", "important_score", "radii", "render"):
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(first["gaussians_count"], exact[0]["gaussians_count"])


def test_bounded_backward_is_bit_identical():
    g, cams, pipe, bg = _scene(N=8000, W=200, H=120, scale=0.05)
    gimg = torch.randn(3, 120, 200, generator=torch.Generator().manual_seed(1)).to(DEV)

    def grads():
        pc = syn.SyntheticGaussians(*[t.detach().clone().requires_grad_(True) for t in
                                      (g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity)], 3, 3)
        out = []
        for c in cams[:3]:
            for t in (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity):
                t.grad = None
            pkg = render(c, pc, pipe, bg)
            (pkg["render"] * gimg).sum().backward()
            out.append([pkg["render"].detach().clone(), pkg["viewspace_points"].grad.clone()] +
                       [t.grad.clone() for t in (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity)])
        return out

    a = grads()
    rasterizer.set_option("sync_free", True)
    render(cams[0], g, pipe, bg)                                    # exact, learns the capacity
    b = grads()
    assert rasterizer.pending_overflow() is False
    for va, vb in zip(a, b):
        for ta, tb in zip(va, vb):
            assert torch.equal(ta, tb)


def test_overflow_is_reported_on_the_device_and_the_view_contributes_zeros():
    g, cams, pipe, bg = _scene()
    with torch.no_grad():
        ref = count_render(cams[1], g, pipe, bg)
        rasterizer.set_option("sync_free", True)
        key = (DEV.index, g.get_xyz.shape[0], 320, 240)
        rasterizer._CAPACITY[key] = 1000                            # far too small
        out = count_render(cams[1], g, pipe, bg)
        assert rasterizer.pending_status() == [True]
        assert int(out["gaussians_count"].sum()) == 0 and float(out["important_score"].abs().sum()) == 0.0
        assert torch.equal(out["radii"], ref["radii"])             # K1 ran; only the binning was abandoned
        assert rasterizer._CAPACITY[key] > 1000                     # raised from the count the device reported
        again = count_render(cams[1], g, pipe, bg)
        assert rasterizer.pending_status() == [False]
        assert torch.equal(again["gaussians_count"], ref["gaussians_count"])
        # a depth beyond max_depth: reported too, and the shape goes back to the exact path
        rasterizer.set_option("max_depth", 1.0)
        out = count_render(cams[1], g, pipe, bg)
        assert rasterizer.pending_status() == [True] and rasterizer._CAPACITY[key] == -1   # this shape stays on the exact path
    # backward through an abandoned view: zero gradients, no out-of-bounds reads
    rasterizer.set_option("max_depth", 100.0)
    rasterizer._CAPACITY[key] = 1000
    rasterizer.set_option("sync_free", True)
    pc = syn.SyntheticGaussians(*[t.detach().clone().requires_grad_(True) for t in
                                  (g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity)], 3, 3)
    pkg = render(cams[1], pc, pipe, bg)
    pkg["render"].sum().backward()
    assert rasterizer.pending_status() == [True]
    assert float(pc._xyz.grad.abs().sum()) == 0.0 and float(pc._features_rest.grad.abs().sum()) == 0.0


def test_sharded_pass_single_thread_sync_free_equals_host_threads_and_the_plain_loop():
    g, cams, pipe, bg = _scene(N=15000, W=256, H=160)
    cams = cams * 3
    with torch.no_grad():
        c0, s0 = prune.prune_list(prune._FrozenGetters(g), list(cams), pipe, bg)
        c1, s1 = prune.prune_list_sharded(g, list(cams), pipe, bg, streams=3, block=5)
        c2, s2 = prune.prune_list_sharded(g, list(cams), pipe, bg, streams=3, block=5, host_threads=True)
        # capacity too small for every view: each block is repaired on the exact path
        rasterizer._CAPACITY[(DEV.index, 15000, 256, 160)] = 64
        c3, s3 = prune.prune_list_sharded(g, list(cams), pipe, bg, streams=3, block=4)
    for c, s in ((c1, s1), (c2, s2), (c3, s3)):
        assert torch.equal(c, c0) and torch.equal(s, s0)


def test_backward_over_views_single_thread_equals_host_threads():
    g, cams, pipe, bg = _scene(N=6000, W=160, H=96, scale=0.05)
    targets = [torch.rand(3, 96, 160, device=DEV) for _ in cams]
    loss_fn = lambda img, gt: (img - gt).abs().mean()  # noqa: E731

    def run(**kw):
        pc = syn.SyntheticGaussians(*[t.detach().clone().requires_grad_(True) for t in
                                      (g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity)], 3, 3)
        losses = parallel.backward_over_views(pc, cams, targets, pipe, bg, loss_fn, streams=3, **kw)
        return [float(x) for x in losses], [t.grad.clone() for t in (pc._xyz, pc._features_rest, pc._scaling, pc._opacity)]

    la, ga = run(host_threads=True)
    lb, gb = run()
    assert la == lb
    for x, y in zip(ga, gb):
        assert torch.equal(x, y)


def test_validated_mode_is_bit_identical_and_repairs_an_overflow_transparently():
    """sync_free="validated": bounded forward whose status words reach the host before the call returns.  Same results as
    the exact path; a view that does not fit is re-run on the exact path inside the same call."""
    g, cams, pipe, bg = _scene()
    with torch.no_grad():
        exact = [count_render(c, g, pipe, bg) for c in cams]
        rasterizer.set_option("sync_free", "validated")
        val = [count_render(c, g, pipe, bg) for c in cams]          # first exact (learns), rest bounded + validated
        key = (DEV.index, g.get_xyz.shape[0], 320, 240)
        assert key in rasterizer._CAPACITY
        rasterizer._CAPACITY[key] = 500                            # every following view overflows once, then is repaired
        rep = [count_render(c, g, pipe, bg) for c in cams]
    assert rasterizer.pending_status() == []                       # validated mode leaves nothing pending
    for a, b, c in zip(exact, val, rep):
        for k in ("gaussians_count", "important_score", "radii", "render"):
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    assert rasterizer._CAPACITY[key] > 500
    # training step through render(): gradients identical to the exact path
    gimg = torch.randn(3, 240, 320, generator=torch.Generator().manual_seed(2)).to(DEV)

    def grads(mode):
        rasterizer.set_option("sync_free", mode)
        pc = syn.SyntheticGaussians(*[t.detach().clone().requires_grad_(True) for t in
                                      (g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity)], 3, 3)
        (render(cams[2], pc, pipe, bg)["render"] * gimg).sum().backward()
        return [t.grad.clone() for t in (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity)]
    for x, y in zip(grads(False), grads("validated")):
        assert torch.equal(x, y)


def test_a_prune_pass_on_one_thread_and_renders_on_another_do_not_disturb_each_other():
    """r2 verdict: prune_list_sharded / _ViewRunner used to flip module-level switches (skip_color_in_count, sync_free) while they
    ran: a render() issued by another thread meanwhile came back as uninitialised memory or took another forward mode.  The
    switches are per call / per thread now: both threads get exactly what they get alone, and the process defaults stay put."""
    import threading
    rasterizer.set_option("sync_free", "validated")                # the shipped default
    g, cams, pipe, bg = _scene(N=40000, W=480, H=320, scale=0.02)
    views = cams * 6
    with torch.no_grad():
        ref_cnt, ref_imp = prune.prune_list_sharded(g, views, pipe, bg, streams=3, block=5)
        ref_img = [render(c, g, pipe, bg)["render"].clone() for c in cams]
        ref_cr = [count_render(c, g, pipe, bg)["render"].clone() for c in cams]
    defaults = dict(rasterizer._OPTIONS)
    out, errors, stop = {}, [], threading.Event()

    def pruner():
        try:
            torch.cuda.set_device(DEV)
            res = []
            with torch.no_grad():
                for _ in range(4):
                    res.append(prune.prune_list_sharded(g, views, pipe, bg, streams=3, block=5))
            out["prune"] = res
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
        finally:
            stop.set()

    t = threading.Thread(target=pruner)
    t.start()
    imgs, crs, n = [], [], 0
    with torch.no_grad():
        while not stop.is_set() or n < len(cams):
            k = n % len(cams)
            imgs.append((k, render(cams[k], g, pipe, bg)["render"]))
            crs.append((k, count_render(cams[k], g, pipe, bg)["render"]))      # a count render that DOES want its image
            assert rasterizer._OPTIONS == defaults                              # nobody switched anything under our feet
            n += 1
            if n > 400:
                break
    t.join()
    assert not errors, errors
    assert n >= len(cams)
    for k, im in imgs:
        assert torch.equal(im, ref_img[k]), f"render() of view {k} changed while a prune pass ran on another thread"
    for k, im in crs:
        assert torch.equal(im, ref_cr[k]), f"count_render() image of view {k} changed while a prune pass ran on another thread"
    for cnt, imp in out["prune"]:
        assert torch.equal(cnt, ref_cnt) and torch.equal(imp, ref_imp)
    assert rasterizer._OPTIONS == defaults and rasterizer.pending_status() == []
