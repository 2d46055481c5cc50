"""CPU, gloo world 2: data-parallel training of an unmodified trainer loop (lightgaussian_amd/dp.py, run.py --distributed).

A stand-in GaussianModel / Scene with the reference's surface (scene/gaussian_model.py:184-217 training_setup with one named
parameter per group, Scene.getTrainCameras, add_densification_stats, densify_and_prune, max_radii2D) and a trainer loop copied in
structure from prune_finetune.py:141-168,287-289 (stack of cameras, pop(randint), render, loss.backward(), optimizer.step(),
zero_grad).  The "renderer" is a differentiable torch function with a per-camera visibility mask, so that the visible-rows
exchange has rows to skip.  Checked:
  * after 3 steps every rank holds the same parameters, bit for bit, and they equal ONE process that renders the same two views
    per step and averages their gradients;
  * the shards are disjoint, the full list stays reachable, rows outside the union were not exchanged;
  * densification statistics and max_radii2D are reduced, a diverged N raises instead of hanging;
  * the launcher path: python -m torch.distributed.run ... -m lightgaussian_amd.run --distributed --backend=gloo on a
    reference-shaped checkout: process group up, cameras sharded, optimizer wrapped, -m redirected for rank 1.
"""
import json
import os
import random
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common  # noqa: F401
from lightgaussian_amd import dp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, NCAM, STEPS = 257, 6, 3
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
SHAPES = {"xyz": (N, 3), "f_dc": (N, 1, 3), "f_rest": (N, 15, 3), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Model:
    """The slice of GaussianModel the trainers and dp.py touch."""

    def __init__(self):
        g = torch.Generator().manual_seed(5)
        for n in NAMES:
            setattr(self, "_" + {"f_dc": "features_dc", "f_rest": "features_rest"}.get(n, n),
                    torch.nn.Parameter(torch.randn(SHAPES[n], generator=g) * 0.1))
        self.xyz_gradient_accum = torch.zeros(N, 1); self.denom = torch.zeros(N, 1); self.max_radii2D = torch.zeros(N)
        self.optimizer = None

    get_xyz = property(lambda self: self._xyz)

    def _params(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    def training_setup(self, training_args):           # scene/gaussian_model.py:184-217: one named group per parameter
        self.optimizer = torch.optim.Adam([{"params": [p], "lr": 0.01, "name": n} for p, n in zip(self._params(), NAMES)], lr=0.0, eps=1e-15)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def densify_and_prune(self, drop, split=False):
        if drop:
            self._xyz = torch.nn.Parameter(self._xyz[:-drop].detach())
        if split:     # scene/gaussian_model.py:666-700 densify_and_split in effect: new positions drawn from the process's RNG stream
            self._xyz = torch.nn.Parameter(torch.cat((self._xyz.detach(), self._xyz.detach()[:7] + torch.normal(torch.zeros(7, 3), 0.01))))


class Scene:
    def __init__(self):
        cams = list(range(NCAM))
        random.Random(3).shuffle(cams)                 # scene/__init__.py:82-88: shuffled once, identically on every rank
        self.train_cameras = {1.0: cams}

    def getTrainCameras(self, scale=1.0):
        return self.train_cameras[scale]


def _visible(model, cam):
    ang = 0.9 * cam
    return (model._xyz.detach()[:, 0] * np.cos(ang) + model._xyz.detach()[:, 2] * np.sin(ang)) > -0.02


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """A differentiable stand-in: every parameter of a VISIBLE Gaussian enters the image, invisible ones do not (zero gradient rows)."""
    vis = _visible(pc, viewpoint_camera)
    w = vis.float().view(-1, 1)
    feat = (pc._xyz * w).sum(1) + (pc._features_dc.view(N, -1) * w).sum(1) + (pc._features_rest.view(N, -1) * w).pow(2).sum(1) \
        + torch.sigmoid(pc._opacity.view(N)) * w.view(N) + (torch.exp(pc._scaling) * w).sum(1) + (pc._rotation * w).pow(2).sum(1)
    image = torch.sin(feat * (1.0 + 0.1 * viewpoint_camera)).view(1, 1, N)
    points = torch.zeros(N, 3, requires_grad=True)
    return {"render": image + 0.0 * points.sum(), "viewspace_points": points, "visibility_filter": vis, "radii": vis.int() * (3 + viewpoint_camera)}


def trainer_loop(model, scene, render_fn, steps, log):
    """prune_finetune.py:141-168,287-289 in structure."""
    from random import randint
    viewpoint_stack = None
    for _ in range(steps):
        if not viewpoint_stack:
            viewpoint_stack = scene.getTrainCameras().copy()
        cam = viewpoint_stack.pop(randint(0, len(viewpoint_stack) - 1))
        log.append(cam)
        pkg = render_fn(cam, model, None, None)
        loss = (pkg["render"] - 0.25).abs().mean()
        loss.backward()
        with torch.no_grad():
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp.install(Model, Scene)
        random.seed(0); torch.manual_seed(0)            # utils/general_utils.py:147-150 safe_state: the same seeds on every rank
        model, scene = Model(), Scene()
        model.training_setup(None)
        assert model.optimizer._lg_dp_wrapped
        mine, full = scene.getTrainCameras(), scene._lg_all_train_cameras()
        assert mine == full[rank::world] and len(full) == NCAM
        from lightgaussian_amd import prune as lg_prune
        assert lg_prune._train_cameras(scene) == full               # the significance pass sees the whole list
        log = []
        dp.configure(check=True)
        trainer_loop(model, scene, dp.wrap_render(render), STEPS, log)
        st = dp.stats()
        assert st["steps"] == STEPS and st["dense_steps"] == 0 and 0 < st["rows_exchanged"] < STEPS * N
        # densification bookkeeping: summed over the ranks
        pts = torch.zeros(N, 3, requires_grad=True)
        pts.grad = torch.full((N, 3), float(rank + 1))
        filt = torch.arange(N) % (rank + 2) == 0
        model.add_densification_stats(pts, filt)
        model.max_radii2D[:] = torch.arange(N).float() * (1 if rank == 0 else -1) + 5 * rank
        model.densify_and_prune(0)
        np.savez(os.path.join(out_dir, f"dp{rank}.npz"), log=np.asarray(log), accum=model.xyz_gradient_accum.numpy(), denom=model.denom.numpy(),
                 radii=model.max_radii2D.numpy(), **{n: p.detach().numpy() for n, p in zip(NAMES, model._params())})
        # a diverged N must raise on every rank, not hang
        try:
            model.densify_and_prune(rank)               # rank 1 drops a Gaussian, rank 0 does not
            raised = False
        except RuntimeError as e:
            raised = "diverged" in str(e)
        assert raised
        # ... and so must replicas that diverged in VALUE with N intact (r5 verdict): both ranks split the same seven Gaussians, but rank 1's
        # random-number stream is one draw ahead (utils/general_utils.py:147-151 seeds every rank alike; anything rank-dependent that draws
        # breaks that silently) -- the positions differ, the count does not
        m3 = Model(); m3.training_setup(None)
        torch.manual_seed(123)
        m3.densify_and_prune(0, split=True)               # same stream on both ranks: passes, N grew by 7 everywhere
        assert m3.get_xyz.shape[0] == N + 7
        torch.manual_seed(123)
        if rank == 1:
            torch.randn(1)
        try:
            m3.densify_and_prune(0, split=True)
            raised = False
        except RuntimeError as e:
            raised = "diverged in VALUE" in str(e)
        assert raised
        # the dense fallback: a step whose renders dp did not see
        m2 = Model(); m2.training_setup(None)
        for p in m2._params():
            p.grad = torch.full_like(p, float(rank))
        m2.optimizer.step()
        assert dp.stats()["dense_steps"] == 1
        ref = Model(); ref.training_setup(None)
        for p in ref._params():
            p.grad = torch.full_like(p, 0.5)
        dp.uninstall()
        ref.optimizer.step()
        for a, b in zip(m2._params(), ref._params()):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def test_two_ranks_train_one_model_equal_to_one_process_averaging_the_same_views(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"dp{r}.npz") for r in range(world)]
    for n in NAMES:
        assert np.array_equal(outs[0][n], outs[1][n]), f"{n} differs between the ranks"
    # the ranks drew from disjoint shards
    full = Scene().getTrainCameras()
    for r in range(world):
        assert set(outs[r]["log"].tolist()) <= set(full[r::world])
    # one process: per step render the two views the ranks drew, average the gradients, step
    model = Model(); model.training_setup(None)
    for s in range(STEPS):
        grads = []
        for r in range(world):
            for p in model._params():
                p.grad = None
            pkg = render(int(outs[r]["log"][s]), model, None, None)
            (pkg["render"] - 0.25).abs().mean().backward()
            grads.append([p.grad.clone() for p in model._params()])
        for p, a, b in zip(model._params(), *grads):
            p.grad = (a + b) / world
        model.optimizer.step()
    for n, p in zip(NAMES, model._params()):
        assert np.array_equal(outs[0][n], p.detach().numpy()), f"{n}: data-parallel result differs from the averaged single process"
    # densification statistics: rank r added r + 1 ... to every (r + 2)-th Gaussian; both ranks hold the sum
    idx = np.arange(N)
    want_acc = (idx % 2 == 0) * np.sqrt(2.0) * 1 + (idx % 3 == 0) * np.sqrt(2.0) * 2
    want_den = (idx % 2 == 0) * 1.0 + (idx % 3 == 0) * 1.0
    for o in outs:
        assert np.allclose(o["accum"].reshape(-1), want_acc, rtol=1e-6) and np.array_equal(o["denom"].reshape(-1), want_den)
        assert np.array_equal(o["radii"], np.maximum(idx, 5 - idx).astype(np.float32))


def test_replica_digest_sees_one_changed_element_and_is_a_device_side_reduction():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 3, generator=g)
    y = x.clone(); y[437, 1] = torch.nextafter(y[437, 1], torch.tensor(10.0))       # one ulp in one coordinate
    a, b = dp.replica_digest(x), dp.replica_digest(y)
    assert a.dtype == torch.int64 and a.dim() == 0 and int(a) == int(dp.replica_digest(x.clone())) and int(a) != int(b)
    assert int(dp.replica_digest(x[:0])) == 0
    assert int(dp.replica_digest(x.t().contiguous().t())) == int(a)                  # values, not strides


def test_shards_cover_every_camera_once_and_degenerate_cases():
    cams = list(range(10))
    assert sorted(sum((dp.shard_cameras(cams, r, 4) for r in range(4)), [])) == cams
    assert dp.shard_cameras([7], 3, 8) == [7]                       # fewer cameras than ranks: the whole list
    assert not dp.active()
    dp.assert_same_count(5)                                          # no process group: nothing to compare
    # world 1 / no group: the hooks are the originals' behaviour
    dp.install(Model, Scene)
    try:
        s = Scene()
        assert s.getTrainCameras() == s._lg_all_train_cameras()
        m = Model(); m.training_setup(None)
        for p in m._params():
            p.grad = torch.ones_like(p)
        m.optimizer.step()                                           # no exchange, plain Adam
        pts = torch.zeros(N, 3, requires_grad=True); pts.grad = torch.ones(N, 3)
        m.add_densification_stats(pts, torch.arange(N) < 4)
        assert float(m.denom.sum()) == 4
    finally:
        dp.uninstall()
    assert not hasattr(Scene, "_lg_all_train_cameras") and not getattr(Model.training_setup, "_lg_dp", False)


def test_model_path_of_the_other_ranks_is_redirected():
    from lightgaussian_amd import run as lg_run
    argv = ["-s", "scene", "-m", "out/run1", "--iterations", "5"]
    assert lg_run._redirect_model_path(argv, 0) == argv
    assert lg_run._redirect_model_path(argv, 3) == ["-s", "scene", "-m", os.path.join("out/run1", ".rank3"), "--iterations", "5"]
    assert lg_run._redirect_model_path(["--model_path=o"], 1) == ["--model_path=" + os.path.join("o", ".rank1")]


def test_distributed_runner_on_a_reference_shaped_checkout(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m lightgaussian_amd.run --distributed --backend=gloo trainer.py -m out:
    the process group is up before the trainer starts, Scene.getTrainCameras is sharded, the optimizer of training_setup()
    averages the gradients in front of step(), rank 1 writes under out/.rank1."""
    root = tmp_path / "LightGaussian"
    (root / "gaussian_renderer").mkdir(parents=True); (root / "utils").mkdir(); (root / "scene").mkdir()
    (root / "gaussian_renderer" / "__init__.py").write_text("def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):\n    return 'literal'\n"
                                                            "def count_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):\n    return 'literal'\n")
    (root / "utils" / "__init__.py").write_text("")
    (root / "utils" / "loss_utils.py").write_text("def l1_loss(network_output, gt):\n    return 'literal'\ndef ssim(img1, img2, window_size=11, size_average=True):\n    return 'literal'\n")
    (root / "prune.py").write_text("def prune_list(gaussians, scene, pipe, background):\n    return 'literal'\ndef calculate_v_imp_score(gaussians, imp_list, v_pow):\n    return 'literal'\n")
    (root / "scene" / "gaussian_model.py").write_text(
        "import torch\n"
        "class GaussianModel:\n"
        "    def __init__(self, sh_degree=3):\n"
        "        self._xyz = torch.nn.Parameter(torch.ones(11, 3))\n"
        "        self._opacity = torch.nn.Parameter(torch.ones(11, 1))\n"
        "    get_xyz = property(lambda self: self._xyz)\n"
        "    def training_setup(self, training_args):\n"
        "        self.optimizer = torch.optim.SGD([{'params': [self._xyz], 'lr': 1.0, 'name': 'xyz'}, {'params': [self._opacity], 'lr': 1.0, 'name': 'opacity'}], lr=0.0)\n"
        "    def prune_points(self, mask):\n        pass\n"
        "    def prune_gaussians(self, percent, import_score):\n        pass\n")
    (root / "scene" / "__init__.py").write_text(
        "from scene.gaussian_model import GaussianModel\n"
        "class Scene:\n"
        "    def __init__(self):\n        self.train_cameras = {1.0: list(range(7))}\n"
        "    def getTrainCameras(self, scale=1.0):\n        return self.train_cameras[scale]\n")
    (root / "trainer.py").write_text(
        "import json, os, sys\n"
        "import torch, torch.distributed as dist\n"
        "from gaussian_renderer import render\n"
        "from scene import Scene, GaussianModel\n"
        "rank = dist.get_rank() if dist.is_initialized() else -1\n"
        "g = GaussianModel(3); s = Scene(); g.training_setup(None)\n"
        "g._xyz.grad = torch.full((11, 3), float(rank)); g._opacity.grad = torch.full((11, 1), 2.0 * rank)\n"
        "g.optimizer.step()\n"
        "rec = dict(rank=rank, world=dist.get_world_size() if dist.is_initialized() else 0, cams=s.getTrainCameras(), all=len(s._lg_all_train_cameras()),\n"
        "           argv=sys.argv[1:], xyz=float(g._xyz[0, 0]), op=float(g._opacity[0, 0]), render=render.__module__, wrapped=hasattr(render, '__wrapped__'))\n"
        "open(os.path.join(os.environ['LG_TEST_OUT'], f'rec{rank}.json'), 'w').write(json.dumps(rec))\n")
    env = dict(os.environ, LG_TEST_OUT=str(tmp_path), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "-m", "lightgaussian_amd.run", "--distributed", "--backend=gloo",
                        str(root / "trainer.py"), "-m", "out"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.load(open(tmp_path / f"rec{k}.json")) for k in range(2)]
    for k, rec in enumerate(recs):
        assert rec["rank"] == k and rec["world"] == 2 and rec["all"] == 7
        assert rec["cams"] == list(range(7))[k::2]
        assert rec["render"] == "lightgaussian_amd.gaussian_renderer" and rec["wrapped"]
        # SGD, lr 1: 1 - mean(rank) = 0.5 and 1 - mean(2 rank) = 0 on both ranks
        assert rec["xyz"] == 0.5 and rec["op"] == 0.0
    assert recs[0]["argv"] == ["-m", "out"] and recs[1]["argv"] == ["-m", os.path.join("out", ".rank1")]
