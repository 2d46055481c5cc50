"""Data-parallel training of the reference's UNMODIFIED trainers (`python -m lightgaussian_amd.run --distributed`).

One process per GPU, Gaussians replicated, every rank draws its cameras from its own shard of the train list, and the gradients
of the step are averaged over the ranks right before `optimizer.step()` -- a global batch of `world` views per optimizer step
where the reference trains on one (prune_finetune.py:144-168,287-289; distill_train.py:124-166; train_densify_prune.py:118-212).
Nothing in the trainers is edited; the glue hangs on four methods of the reference's classes:

    Scene.getTrainCameras                    rank r sees cameras r, r + world, ... of the (identically shuffled, scene/__init__.py:82-88)
                                             list: the trainers' `viewpoint_stack = scene.getTrainCameras().copy()` then walks a
                                             disjoint shard per rank -- an epoch still visits every camera exactly once.  The full
                                             list stays reachable as scene._lg_all_train_cameras() (the significance pass needs it:
                                             every rank must hand prune_list_sharded the same sequence).
    GaussianModel.training_setup             the optimizer it creates (scene/gaussian_model.py:184-217) gets its step() wrapped:
                                             average the `.grad` of the six parameter groups over the ranks, then step.  Rows no
                                             rank's camera saw are exactly zero everywhere and are not exchanged
                                             (parallel.allreduce_gradients_visible: visibility flags MAX-reduced + ONE packed sum all-reduce -- few,
                                             large collectives for point-to-point xGMI); the visibility comes from the render()
                                             calls of the step (note_render, installed around the patched gaussian_renderer.render).
    GaussianModel.add_densification_stats    (train_densify_prune.py:175) the per-view statistics are summed over the ranks, so
                                             xyz_gradient_accum / denom -- and with them every densification decision -- are the same
                                             on every rank.
    GaussianModel.densify_and_prune          max_radii2D is max-reduced first (the trainer updates it in its own loop,
                                             train_densify_prune.py:172-174, from this rank's view only); afterwards the ranks check
                                             that they still hold the same number of Gaussians.

Consistency rule: every rank must execute the same collectives in the same order with the same N.  The parameters start equal
(same checkpoint / point cloud, same seeds: utils/general_utils.py:147-151), receive the same averaged gradients, and every
pruning / densification decision is a deterministic function of all-reduced data, so they stay equal; `assert_same_count()`
verifies it (one 8-byte all-gather) wherever N can change.  A mismatch raises on every rank instead of hanging in the next
collective.  Output files: ranks other than 0 write under <model_path>/.rank<r> (run.py rewrites their -m argument).
"""
import os
import threading

import torch
import torch.distributed as dist

from . import parallel

_STATE = {"installed": [], "group": None, "visible": {}, "lock": threading.Lock(), "steps": 0, "rows": 0, "dense_steps": 0}


def active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(_STATE["group"]) > 1


def _rank_world():
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(_STATE["group"]), dist.get_world_size(_STATE["group"])


def assert_same_count(n, what="Gaussians"):
    """Every rank must hold the same number of Gaussians before a collective sized by it.  One int64 all-gather."""
    if not active():
        return
    rank, world = _rank_world()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(_STATE["group"]) == "nccl" else torch.device("cpu")
    mine = torch.tensor([int(n)], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine, group=_STATE["group"])
    counts = [int(t.item()) for t in allc]
    if len(set(counts)) != 1:
        raise RuntimeError(f"data-parallel ranks diverged: number of {what} per rank = {counts} (rank {rank}); the next collective would "
                           "mismatch.  Every rank must load the same model and take the same prune / densify decisions.")


# ---- visibility of the step's renders -------------------------------------------------------------------------------------------

def note_render(pc, pkg):
    """Called with the result of every render() of a step: remembers (ORs) which Gaussians this rank's view(s) saw, keyed by the
    model's parameter tensor, for the gradient exchange in front of the next optimizer.step()."""
    if not torch.is_grad_enabled():
        return
    xyz = getattr(pc, "_xyz", None)
    vis = pkg.get("visibility_filter") if isinstance(pkg, dict) else None
    if xyz is None or vis is None or not getattr(xyz, "requires_grad", False):
        return
    with _STATE["lock"]:
        key = id(xyz)
        prev = _STATE["visible"].pop(key, None)
        # (clone: the fused path returns a view into the forward's saved geom buffer)
        _STATE["visible"][key] = (xyz, vis.detach().clone() if prev is None or prev[1].shape != vis.shape else prev[1].logical_or_(vis.detach()))
        while len(_STATE["visible"]) > 8:                 # models that are rendered with grad but never stepped: bounded
            _STATE["visible"].pop(next(iter(_STATE["visible"])))


def wrap_render(render_fn):
    """render() that also records the visibility for the gradient exchange (same signature, same result)."""
    def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, **kw):
        pkg = render_fn(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, **kw)
        note_render(pc, pkg)
        return pkg
    render.__wrapped__ = render_fn
    render.__module__ = getattr(render_fn, "__module__", __name__)
    render.__qualname__ = getattr(render_fn, "__qualname__", "render")
    render.__doc__ = render_fn.__doc__
    return render


def _take_visible(params):
    """The recorded visibility whose model owns `params` (by the identity of its _xyz tensor), or None."""
    ids = {id(p) for p in params}
    with _STATE["lock"]:
        hit = None
        for key, (xyz, vis) in list(_STATE["visible"].items()):
            if key in ids and any(xyz is p for p in params):
                hit = vis
                del _STATE["visible"][key]
    return hit


def exchange_gradients(optimizer, check=None, force=False):
    """Average the gradients of every parameter group of `optimizer` over the ranks (in place).  Visible-rows exchange when the
    step's renders were seen by note_render (all six tensors have one row per Gaussian), the dense bucketed all-reduce otherwise.
    check (default: env LG_DP_CHECK=1): verify that rows outside this rank's visibility are exactly zero -- the precondition of
    the visible-rows exchange; it holds for the reference's photometric losses, not for a regulariser that touches unseen rows.
    force: exchange at world size 1 too (an initialised group is still required): the RCCL code path on a 1-GPU box."""
    if not (active() or (force and dist.is_available() and dist.is_initialized())):
        return None
    params = [p for g in optimizer.param_groups for p in g["params"]]
    missing = [i for i, p in enumerate(params) if p.grad is None]
    if missing and len(missing) != len(params):
        raise RuntimeError(f"data-parallel step: parameter groups {missing} have no gradient on this rank while others do; every rank must "
                           "differentiate the same set of parameters (the flat exchange buffers would differ in size)")
    if missing:
        return None
    vis = _take_visible(params)
    N = params[0].shape[0]
    rows_ok = vis is not None and vis.shape[0] == N and all(p.grad.shape[0] == N for p in params)
    if check is None:
        check = os.environ.get("LG_DP_CHECK", "0") == "1"
    _STATE["steps"] += 1
    if rows_ok:
        if check:
            hidden = ~vis.reshape(-1).bool()
            for p in params:
                if bool((p.grad.reshape(N, -1)[hidden] != 0).any()):
                    raise RuntimeError("data-parallel step: a gradient row of a Gaussian no render of this step saw is non-zero; the "
                                       "visible-rows exchange would leave it unreduced (use LG_DP_DENSE=1)")
        if os.environ.get("LG_DP_DENSE", "0") != "1":
            k, _ = parallel.allreduce_gradients_visible(params, vis, group=_STATE["group"], force=force)
            _STATE["rows"] += k
            return {"rows": k, "of": N, "mode": "visible"}
    _STATE["dense_steps"] += 1
    n = parallel.allreduce_gradients(params, group=_STATE["group"], force=force)
    return {"collectives": n, "of": N, "mode": "dense"}


def wrap_optimizer(optimizer):
    """optimizer.step() := exchange_gradients(optimizer); step().  Idempotent."""
    if getattr(optimizer, "_lg_dp_wrapped", False):
        return optimizer
    inner = optimizer.step

    def step(*a, **kw):
        # (LG_DP_FORCE=1: exchange at world size 1 too -- the RCCL code path of a 1-GPU box, tests/test_gpu_dp_runner.py)
        exchange_gradients(optimizer, force=os.environ.get("LG_DP_FORCE", "0") == "1")
        return inner(*a, **kw)

    optimizer.step = step
    optimizer._lg_dp_wrapped = True
    return optimizer


# ---- patches on the reference's classes ------------------------------------------------------------------------------------------

def _set(owner, name, new):
    _STATE["installed"].append((owner, name, owner.__dict__.get(name, None), name in owner.__dict__))
    setattr(owner, name, new)


def shard_cameras(cams, rank, world):
    """Cameras of `rank`: every world-th of the list (an empty shard -- fewer cameras than ranks -- falls back to the whole list)."""
    mine = list(cams)[rank::world]
    return mine if mine else list(cams)


def install(gaussian_model_cls=None, scene_cls=None, group=None):
    """Hang the data-parallel glue on the reference's classes (see the module docstring).  Safe to call at world size 1 (every hook
    degenerates to the original behaviour).  uninstall() restores the classes."""
    _STATE["group"] = group
    if scene_cls is not None and hasattr(scene_cls, "getTrainCameras") and not hasattr(scene_cls, "_lg_all_train_cameras"):
        orig_get = scene_cls.getTrainCameras

        def getTrainCameras(self, scale=1.0):
            rank, world = _rank_world()
            full = orig_get(self, scale)
            return shard_cameras(full, rank, world) if world > 1 else full

        _set(scene_cls, "_lg_all_train_cameras", orig_get)
        _set(scene_cls, "getTrainCameras", getTrainCameras)
    gm = gaussian_model_cls
    if gm is not None and hasattr(gm, "training_setup") and not getattr(gm.training_setup, "_lg_dp", False):
        orig_setup = gm.training_setup

        def training_setup(self, training_args):
            out = orig_setup(self, training_args)
            wrap_optimizer(self.optimizer)
            return out

        training_setup._lg_dp = True
        _set(gm, "training_setup", training_setup)
    if gm is not None and hasattr(gm, "add_densification_stats") and not getattr(gm.add_densification_stats, "_lg_dp", False):
        def add_densification_stats(self, viewspace_point_tensor, update_filter):
            """scene/gaussian_model.py:784-788 with the per-view increments summed over the ranks."""
            if not active():
                self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
                self.denom[update_filter] += 1
                return
            n = self.xyz_gradient_accum.shape[0]
            inc = torch.zeros((n, 2), dtype=self.xyz_gradient_accum.dtype, device=self.xyz_gradient_accum.device)
            inc[update_filter, 0:1] = torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
            inc[update_filter, 1] = 1
            dist.all_reduce(inc, op=dist.ReduceOp.SUM, group=_STATE["group"])
            self.xyz_gradient_accum += inc[:, 0:1]
            self.denom += inc[:, 1:2]

        add_densification_stats._lg_dp = True
        _set(gm, "add_densification_stats", add_densification_stats)
    if gm is not None and hasattr(gm, "densify_and_prune") and not getattr(gm.densify_and_prune, "_lg_dp", False):
        orig_dap = gm.densify_and_prune

        def densify_and_prune(self, *a, **kw):
            if active():
                assert_same_count(self.get_xyz.shape[0])
                dist.all_reduce(self.max_radii2D, op=dist.ReduceOp.MAX, group=_STATE["group"])
            out = orig_dap(self, *a, **kw)
            if active():
                assert_same_count(self.get_xyz.shape[0], "Gaussians after densify_and_prune")
            return out

        densify_and_prune._lg_dp = True
        _set(gm, "densify_and_prune", densify_and_prune)


def uninstall():
    while _STATE["installed"]:
        owner, name, old, had = _STATE["installed"].pop()
        if had:
            setattr(owner, name, old)
        else:
            try:
                delattr(owner, name)
            except AttributeError:
                pass
    _STATE["visible"].clear()
    _STATE["group"] = None


def stats():
    """Counters of the exchanges so far: steps, rows exchanged (visible mode), steps that fell back to the dense all-reduce."""
    return {"steps": _STATE["steps"], "rows_exchanged": _STATE["rows"], "dense_steps": _STATE["dense_steps"]}
