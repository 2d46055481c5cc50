"""CPU, gloo world 2: the round-5 data-parallel exchange (lightgaussian_amd/dp.py + parallel.RankOneSHExchange).

The SH-coefficient gradient of one view is the outer product basis(dir) (x) dRGB; the ranks all-gather dRGB [N, 3] and the camera
centre instead of all-reducing [N, 16, 3], and rebuild the sum locally.  A stand-in render with the rasterizer's contract (option
`sh_grad_sink`: the backward hands dL/d(rgb) to the sink and returns None for the coefficient gradients) drives the unmodified
trainer loop of tests/test_dp_trainer.py.  Checked:
  * three Adam steps with the rank-one exchange leave the parameters of the dense exchange (`dp.configure(sh="dense")`), bit for bit, on both ranks;
  * a camera batch of two views per rank and step (gradient accumulation): equal to 1e-6 (another summation order), ranks bit-equal;
  * a parameter group without a gradient on every rank (train_densify_prune.py:194-197: reset_opacity() between backward() and
    step()) is skipped like single-process Adam skips it -- no raise, ranks stay equal; a group missing on ONE rank raises;
  * bytes on the wire: 12 B per Gaussian and view for the SH part instead of 192.
The HIP kernel behind sh_grad_from_rgb (lg_sh_grad_from_rgb) is pinned against K9's own rows in tests/test_gpu_dp_rank1.py; here the
CPU restatement in parallel.py (checked against autograd through sh_utils.eval_sh below) stands in for it."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common  # noqa: F401
from lightgaussian_amd import dp, parallel, sh_utils

N, NCAM, M, DEG = 193, 8, 16, 3
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
SHAPES = {"xyz": (N, 3), "f_dc": (N, 1, 3), "f_rest": (N, M - 1, 3), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Model:
    def __init__(self):
        g = torch.Generator().manual_seed(11)
        for n in NAMES:
            setattr(self, "_" + {"f_dc": "features_dc", "f_rest": "features_rest"}.get(n, n), torch.nn.Parameter(torch.randn(SHAPES[n], generator=g) * 0.2))
        self.optimizer = None

    get_xyz = property(lambda self: self._xyz)

    def _params(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    def training_setup(self, training_args):
        self.optimizer = torch.optim.Adam([{"params": [p], "lr": 0.01, "name": n} for p, n in zip(self._params(), NAMES)], lr=0.0, eps=1e-15)

    def reset_opacity(self):
        """scene/gaussian_model.py:219-222 in effect: the opacity Parameter is REPLACED (replace_tensor_to_optimizer), its .grad is gone."""
        new = torch.nn.Parameter(torch.clamp(self._opacity.detach(), max=-0.1))
        for g in self.optimizer.param_groups:
            if g["name"] == "opacity":
                st = self.optimizer.state.pop(g["params"][0], None)
                g["params"][0] = new
                if st is not None:
                    self.optimizer.state[new] = st
        self._opacity = new


class Scene:
    def __init__(self):
        cams = list(range(NCAM)); random.Random(3).shuffle(cams)
        self.train_cameras = {1.0: cams}

    def getTrainCameras(self, scale=1.0):
        return self.train_cameras[scale]


def _campos(cam):
    return torch.tensor([3.0 * np.cos(0.7 * cam), 0.3 * cam - 1.0, 3.0 * np.sin(0.7 * cam)], dtype=torch.float32)


class _SHColor(torch.autograd.Function):
    """colours from SH with the rasterizer's gradient contract: with a sink, dL/d(rgb) goes to the sink and the coefficients get None."""

    @staticmethod
    def forward(ctx, xyz, dc, rest, campos, sink):
        d = xyz.detach() - campos
        basis = parallel._sh_basis_rows(d / d.norm(dim=1, keepdim=True), DEG)
        sh = torch.cat((dc, rest), dim=1)
        ctx.save_for_backward(basis)
        ctx.sink, ctx.campos = sink, campos
        return (basis.unsqueeze(-1) * sh).sum(1) + 0.5

    @staticmethod
    def backward(ctx, g):
        (basis,) = ctx.saved_tensors
        if ctx.sink is not None:
            ctx.sink.add(g.contiguous(), ctx.campos, DEG)
            return None, None, None, None, None
        full = basis.unsqueeze(-1) * g.unsqueeze(1)
        return None, full[:, :1], full[:, 1:], None, None


class _Geom(torch.autograd.Function):
    """the geometric parameters with the rasterizer's OTHER contract: when a gradient-chunk hook is installed (rasterizer.set_grad_chunk_hook,
    i.e. parallel.OverlappedGradAllReduce), the backward reports its gradient tensors range by range, as lg_backward_chunked does."""

    @staticmethod
    def forward(ctx, xyz, scaling, rotation, w):
        ctx.save_for_backward(scaling, rotation, w)
        return (xyz * w).sum(1) + (torch.exp(scaling) * w).sum(1) + (rotation * w).pow(2).sum(1)

    @staticmethod
    def backward(ctx, g):
        from lightgaussian_amd import rasterizer
        scaling, rotation, w = ctx.saved_tensors
        g = g.unsqueeze(1)
        g_xyz = (g * w).expand(-1, 3).contiguous()
        g_sc = g * torch.exp(scaling) * w
        g_rot = g * 2.0 * rotation * w * w
        hook = rasterizer._GRAD_CHUNKS["hook"]
        if hook is not None:
            grads = {"_xyz": g_xyz, "_scaling": g_sc, "_rotation": g_rot}
            for first in range(0, N, 80):
                hook(first, min(80, N - first), grads)
        return g_xyz, g_sc, g_rot, None


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, *, options=None):
    cam = int(viewpoint_camera)
    campos = _campos(cam)
    vis = (pc._xyz.detach() @ campos) > -0.3
    w = vis.float().view(-1, 1)
    rgb = _SHColor.apply(pc._xyz, pc._features_dc, pc._features_rest, campos, (options or {}).get("sh_grad_sink"))
    feat = (rgb * w).sum(1) * torch.sigmoid(pc._opacity.view(N)) + _Geom.apply(pc._xyz, pc._scaling, pc._rotation, w)
    image = torch.sin(feat * (1.0 + 0.1 * cam)).view(1, 1, N)
    points = torch.zeros(N, 3, requires_grad=True)
    return {"render": image + 0.0 * points.sum(), "viewspace_points": points, "visibility_filter": vis, "radii": vis.int() * (3 + cam)}


def trainer_loop(model, scene, render_fn, steps, views_per_step=1, reset_at=None, sh_reg=False):
    from random import randint
    stack = None
    for it in range(steps):
        for _ in range(views_per_step):
            if not stack:
                stack = scene.getTrainCameras().copy()
            cam = stack.pop(randint(0, len(stack) - 1))
            pkg = render_fn(cam, model, None, None)
            loss = (pkg["render"] - 0.25).abs().mean()
            if sh_reg:     # a regulariser on the SH coefficients: autograd puts a gradient on f_dc / f_rest NEXT to the sink's dRGB
                loss = loss + 1e-2 * (model._features_rest * (1.0 + 0.1 * cam)).pow(2).mean() + 1e-2 * model._features_dc.abs().mean()
            loss.backward()
        if reset_at == it:
            model.reset_opacity()                          # between backward() and step(), as train_densify_prune.py:194-197
        with torch.no_grad():
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)


def _run(mode, views_per_step=1, reset_at=None, overlap=False, sh_reg=False):
    dp.uninstall()                                     # (also puts every switch back to its default)
    # check_set=True: compare the parameter sets on every step (default: first steps, every 64th, on a change)
    dp.install(Model, Scene, overlap=overlap, sh=mode, check=False, check_set=True)
    random.seed(0); torch.manual_seed(0)
    model, scene = Model(), Scene()
    model.training_setup(None)
    before = dp.stats()
    trainer_loop(model, scene, dp.wrap_render(render), 3, views_per_step, reset_at, sh_reg)
    after = dp.stats()
    return model, {k: after[k] - before[k] for k in after}


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp.install(Model, Scene)
        res = {}
        for tag, kw in (("one", {}), ("two", {"views_per_step": 2}), ("reset", {"reset_at": 1})):
            a, sa = _run("rank1", **kw)
            b, sb = _run("dense", **kw)
            assert sa["rank1_sh_steps"] == 3 and sb["rank1_sh_steps"] == 0
            for n, p, q in zip(NAMES, a._params(), b._params()):
                # one view per rank: (t0 + t1) / 2 either way -- the same bits.  A camera batch: the dense path adds rank-local sums
                # ((a0 + a1) + (b0 + b1)), the rank-one path adds the views one by one (((a0 + b0) + a1) + b1): float rounding apart
                same = torch.equal(p, q) if tag != "two" else torch.allclose(p, q, rtol=1e-6, atol=1e-7)
                assert same, f"{tag}: {n} differs between the rank-one and the dense exchange on rank {rank}"
            res[tag] = [p.detach().numpy() for p in a._params()]
            if tag != "two":
                # the non-SH gradients all-reduced range by range from inside the backward (one backward per step): the same bits, and the
                # opacity group -- which the hook never sees here -- still goes through the dense all-reduce
                c, sc = _run("rank1", overlap=True, **kw)
                assert sc["overlapped_steps"] == 3 and sa["overlapped_steps"] == 0
                for n, p, q in zip(NAMES, a._params(), c._params()):
                    assert torch.equal(p, q), f"{tag}: {n} differs with the overlapped all-reduce on rank {rank}"
            # wire bytes of the SH part: (3 N + 3) floats per view sent once, received from the other rank
            if tag != "reset":
                k = kw.get("views_per_step", 1)
                assert sa["bytes_on_wire"] < 0.5 * sb["bytes_on_wire"]
                assert sa["sh_bytes_on_wire"] == 3 * k * (3 * N + 3) * 4 * world and sb["sh_bytes_on_wire"] == 0
        # an SH group that already holds a gradient of its own (a regulariser on the coefficients) when the sink is finished --
        # that gradient is all-reduced and the rebuilt one ADDED (it used to be overwritten, unreduced): equal to the dense exchange up to
        # the association of the sum, and the same bits on both ranks (the caller compares the files)
        a, sa = _run("rank1", sh_reg=True)
        b, sb = _run("dense", sh_reg=True)
        assert sa["rank1_sh_steps"] == 3 and sb["rank1_sh_steps"] == 0
        for n, p, q in zip(NAMES, a._params(), b._params()):
            assert torch.allclose(p, q, rtol=2e-5, atol=1e-6), f"sh_reg: {n} differs between the rank-one and the dense exchange on rank {rank}"
        plain, _ = _run("rank1")
        assert not torch.allclose(a._features_rest, plain._features_rest, rtol=1e-4, atol=1e-6)      # the regulariser did act
        res["shreg"] = [p.detach().numpy() for p in a._params()]
        # a model whose optimizer is NOT the wrapped one gets no sink (nobody would rebuild its SH gradients), and what was recorded for
        # parameters that a prune / densify replaced before the step is dropped at that step, whatever the size of the store
        dp.uninstall(); dp.install(Model, Scene, sh="rank1", check_set=True)
        loose = Model()
        loose.optimizer = torch.optim.Adam([{"params": [p], "lr": 0.01, "name": n} for p, n in zip(loose._params(), NAMES)], lr=0.0, eps=1e-15)
        assert dp._sink_for(loose, None, None) is None
        wrapped = Model(); wrapped.training_setup(None)
        sink = dp._sink_for(wrapped, None, None)
        assert sink is not None and dp._sink_for(wrapped, None, None) is sink
        pkg = dp.wrap_render(render)(0, wrapped, None, None)
        (pkg["render"] - 0.25).abs().mean().backward()
        assert len(sink.views) == 1 and len(dp._STATE["sinks"]) == 1 and len(dp._STATE["visible"]) == 1
        old_xyz = wrapped._xyz
        wrapped._xyz = torch.nn.Parameter(old_xyz.detach().clone())          # prune_points / densification_postfix: every Parameter is new
        wrapped.optimizer.param_groups[0]["params"][0] = wrapped._xyz
        for q in wrapped._params():
            q.grad = torch.zeros_like(q)
        wrapped.optimizer.step()
        assert not dp._STATE["sinks"] and not dp._STATE["visible"] and sink.views == [] and id(old_xyz) not in dp._STATE["stepped"]
        assert dp._sink_for(wrapped, None, None) is not None                   # the new _xyz is the registered one now
        # a group without a gradient on ONE rank only: refused on every rank (the union / intersection of the masks differ from somebody's)
        m = Model(); m.training_setup(None)
        for i, p in enumerate(m._params()):
            p.grad = None if (i == 3 and rank == 1) else torch.ones_like(p)
        try:
            m.optimizer.step(); raised = False
        except RuntimeError as e:
            raised = "different parameter groups" in str(e)
        assert raised
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **{f"{t}_{n}": v for t, vs in res.items() for n, v in zip(NAMES, vs)})
        dp.uninstall()
    finally:
        dist.destroy_process_group()


def test_rank_one_sh_exchange_equals_the_dense_exchange_bit_for_bit(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for key in outs[0].files:
        assert np.array_equal(outs[0][key], outs[1][key]), f"{key} differs between the ranks"
    # the parameters moved, the SH ones included
    ref = Model()
    assert not np.array_equal(outs[0]["one_f_rest"], ref._features_rest.detach().numpy())
    # reset_opacity: the opacity group was skipped by the step it had no gradient in, not frozen for good
    assert not np.array_equal(outs[0]["reset_opacity"], torch.clamp(ref._opacity.detach(), max=-0.1).numpy())


def test_the_cpu_restatement_of_sh_grad_from_rgb_is_the_gradient_of_eval_sh():
    """parallel.sh_grad_from_rgb on CPU tensors (the gloo tests' stand-in for lg_sh_grad_from_rgb) against autograd through the
    repository's eval_sh (pinned on the reference's utils/sh_utils.py golden vectors), every degree, three views, with accumulate."""
    g = torch.Generator().manual_seed(2)
    xyz = torch.randn(70, 3, generator=g); cams = torch.randn(3, 3, generator=g) * 3
    drgb = torch.randn(3, 70, 3, generator=g)
    for deg in range(4):
        for Mst in {(deg + 1) ** 2, 16}:
            sh = torch.randn(70, Mst, 3, generator=g, requires_grad=True)
            tot = 0
            for v in range(3):
                d = xyz - cams[v]
                tot = tot + (sh_utils.eval_sh(deg, sh.transpose(1, 2), d / d.norm(dim=1, keepdim=True)) * drgb[v]).sum()
            tot.backward()
            gd, gr = parallel.sh_grad_from_rgb(xyz, cams, drgb, deg, Mst, divisor=2.0)
            got = torch.cat((gd, gr), 1)
            assert torch.allclose(got, sh.grad / 2.0, rtol=1e-6, atol=1e-7), (deg, Mst)
            assert float(got[:, (deg + 1) ** 2:].abs().max() if Mst > (deg + 1) ** 2 else 0.0) == 0.0      # beyond the active degree: exactly zero
            # two calls with accumulate == one call
            out = parallel.sh_grad_from_rgb(xyz, cams[:2], drgb[:2], deg, Mst, divisor=1.0)
            out = parallel.sh_grad_from_rgb(xyz, cams[2:], drgb[2:], deg, Mst, divisor=2.0, out=out, accumulate=True)
            assert torch.equal(torch.cat(out, 1), got), (deg, Mst)


def test_switches_are_configuration_not_environment(monkeypatch):
    """The exchange's switches live in dp.configure(); the environment is read once by the launcher (dp.config_from_env), never on the
    step's path: a stray LG_DP_* variable in a shell changes nothing for a process that did not ask for it."""
    dp.uninstall()
    monkeypatch.setenv("LG_DP_SH", "dense")
    monkeypatch.setenv("LG_DP_FORCE", "1")
    assert dp.sh_mode() == "rank1" and not dp._CONFIG["force"]
    cfg = dp.config_from_env()                                   # what `python -m lightgaussian_amd.run` does at start-up
    assert cfg["sh"] == "dense" and cfg["force"] is True and cfg["check_set"] is None and dp.sh_mode() == "dense"
    assert dp.config_from_env({"LG_DP_CHECK_SET": "0", "LG_DP_CHECK": "1", "LG_DP_OVERLAP": "1"})["check_set"] is False
    assert dp._CONFIG["check"] is True and dp._CONFIG["overlap"] is True
    with pytest.raises(TypeError):
        dp.configure(shh="dense")
    with pytest.raises(ValueError):
        dp.configure(sh="sparse")
    dp.uninstall()                                               # back to the defaults
    assert dp._CONFIG == dp._DEFAULTS and dp.sh_mode() == "rank1"
