"""Drop-in runner: the reference's UNMODIFIED trainers on the MI355X path.

    python -m lightgaussian_amd.run /path/to/LightGaussian/prune_finetune.py -s <scene> -m <out> --start_checkpoint ...
    python -m lightgaussian_amd.run /path/to/LightGaussian/distill_train.py ...
    python -m lightgaussian_amd.run /path/to/LightGaussian/train_densify_prune.py ...
    python -m torch.distributed.run --nproc-per-node 8 -m lightgaussian_amd.run --distributed /path/to/prune_finetune.py ...
        (--distributed = DATA-PARALLEL training of the unmodified trainer, lightgaussian_amd/dp.py: every rank draws its cameras
         from its own shard of the train list, gradients are averaged over the ranks in front of every optimizer.step() (rows no
         rank saw are not exchanged), add_densification_stats / max_radii2D are reduced, prune_list is sharded by camera, ranks
         other than 0 write under <model_path>/.rank<r>)
    python -m lightgaussian_amd.run --weight-policy=alpha_t /path/to/prune_finetune.py ...
        (what one (pixel, Gaussian) hit adds to important_score in every count_render of the run: opacity (default, the paper's sigma_j),
         one, alpha or alpha_t -- the weight of the reference's un-vendored fork is not verifiable from its repository, SURVEY section 2.2;
         all four are deterministic and bit-pinned)
    python -m lightgaussian_amd.run --fused-adam /path/to/prune_finetune.py ...
        (opt-in, outside the replaced path: torch.optim.Adam as the trainers construct it, but with fused=True -- fused_adam() below)
    python -m lightgaussian_amd.run --lazy-loss /path/to/prune_finetune.py ...
        (opt-in: l1_loss() / ssim() return lazy scalars, loss_utils.LazyLoss -- the trainers' `(1 - lambda) * Ll1 + lambda * (1 - ssim)`
         line launches nothing, backward() feeds the two coefficients to the fused loss node, loss.item() reads a pinned copy;
         the trainers' iter_start.elapsed_time(iter_end) then waits for its end event itself -- event_timing() below)
    python -m lightgaussian_amd.run --no-iter-timing /path/to/prune_finetune.py ...
        (implies --lazy-loss; elapsed_time of an unfinished pair returns NaN instead of waiting: the host runs ahead of the device)

Why a runner.  The trainers import their collaborators by name from their own directory, which Python puts first on sys.path:
    from utils.loss_utils import l1_loss, ssim                 prune_finetune.py:15, distill_train.py:15, train_densify_prune.py
    from gaussian_renderer import render, count_render         prune_finetune.py:17, distill_train.py:16, train_densify_prune.py:11
    from prune import prune_list, calculate_v_imp_score        prune_finetune.py:39
    from scene import Scene, GaussianModel                     prune_finetune.py:19
With only PYTHONPATH=<this repo> the `diff_gaussian_rasterization` / `simple_knn` shims resolve, i.e. the rasterizer and the
kNN are replaced, but everything around them is still the reference's torch code: render() evaluates the getters in torch on
every call (the literal pattern: 378 views/s instead of 664 at 3 M Gaussians / 1080p, round 4), SSIM is five grouped conv2d (10.8 ms per
step at 1080p, three times the whole render), prune_list walks the views one by one, prune_points runs 21 boolean-index kernels,
the VecTree search materialises cdist.  patch_reference() imports the reference's modules and REBINDS those symbols -- in
the modules themselves and in every module that already imported them by name -- to the implementations of this package:

    reference symbol                                   rebound to
    gaussian_renderer.render / count_render            lightgaussian_amd.gaussian_renderer.render / count_render  (getters fused into K1/K9)
    utils.loss_utils.l1_loss / ssim                    lightgaussian_amd.loss_utils.l1_loss / ssim                (one fused L1 + SSIM launch pair)
    prune.prune_list                                   lightgaussian_amd.prune.prune_list_sharded                 (views in flight; RCCL when a group exists)
    prune.calculate_v_imp_score                        prune_epilogue's v_list (radix select instead of a full sort; bit-identical)
    scene.gaussian_model.GaussianModel.prune_points    lightgaussian_amd.prune.prune_points                       (one compaction launch)
    scene.gaussian_model.GaussianModel.prune_gaussians lightgaussian_amd.prune.prune_gaussians                    (radix select + the above)
    vectree.vq: -torch.cdist(x, c) -> argmax           lightgaussian_amd.vq.nearest_code                          (both sites: EuclideanCodebook.forward
                                                                                                                   :262-266 and kmeans() :131-137)
Same signatures, same return values (tests/test_dropin_runner.py asserts the call contracts on the reference's own modules;
tests/test_gpu_dropin_runner.py runs the body of prune_finetune.py:150-170 through the patched names against the unpatched
literal path).  unpatch_reference() restores everything.  There is no CPU fallback behind any of these: CPU tensors raise.
"""
import importlib
import os
import runpy
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATCHED = []          # (owner object, attribute name, original value) in patch order
_REPORT = {}
_DP_RENDER = {}        # the visibility-recording render() of a data-parallel run (one object, so that patching stays idempotent)


def _rebind_everywhere(old, new):
    """`from module import name` copies the binding: every module that already holds the original under any name gets the
    replacement too (the trainers' own globals included, when patching happens after their imports)."""
    n = 0
    for mod in list(sys.modules.values()):
        d = getattr(mod, "__dict__", None)
        if not isinstance(d, dict) or getattr(mod, "__name__", "").startswith("lightgaussian_amd"):
            continue
        for k, v in list(d.items()):
            if v is old and v is not new:
                _PATCHED.append((mod, k, old))
                setattr(mod, k, new)
                n += 1
    return n


def _set(owner, name, new, label):
    old = getattr(owner, name)
    if old is new:
        return
    _PATCHED.append((owner, name, old))
    setattr(owner, name, new)
    _REPORT[label] = {"old": f"{getattr(old, '__module__', '?')}.{getattr(old, '__qualname__', name)}",
                      "new": f"{getattr(new, '__module__', '?')}.{getattr(new, '__qualname__', name)}"}
    if callable(old) and not isinstance(owner, type):
        _REPORT[label]["also_rebound_in"] = _rebind_everywhere(old, new)


def _module(name):
    try:
        return sys.modules.get(name) or importlib.import_module(name)
    except Exception as e:  # noqa: BLE001 -- a missing optional collaborator (vectree needs einops, scene needs plyfile ...) is reported
        _REPORT[name] = {"skipped": f"{type(e).__name__}: {e}"[:200]}
        return None


# ---- replacements that need an adapter (same signature as the reference symbol they stand in for) -------------------------------

def _prune_list(gaussians, scene, pipe, background):
    """prune.py:133-157 prune_list(gaussians, scene, pipe, background) -> (gaussian_list, imp_list)."""
    import torch
    from . import prune as lg_prune
    from . import dp
    dp.assert_same_count(gaussians.get_xyz.shape[0], values=gaussians.get_xyz)     # data-parallel run: the collectives below are sized by N
    with torch.no_grad():       # (the reference's first view is not detached, prune.py:137-141; nothing downstream differentiates it)
        return lg_prune.prune_list_sharded(gaussians, scene, pipe, background)


def _calculate_v_imp_score(gaussians, imp_list, v_pow):
    """prune.py:112-128; the kth volume by one radix select instead of a full descending sort -- same element, same v_list."""
    import torch
    from . import prune as lg_prune
    if not (torch.is_tensor(imp_list) and imp_list.is_cuda and imp_list.dtype == torch.float32 and imp_list.numel() > 0):
        return lg_prune.calculate_v_imp_score(gaussians, imp_list, v_pow)
    volume = torch.prod(gaussians.get_scaling, dim=1)                                  # prune.py:120
    n = volume.shape[0]
    with torch.no_grad():                                                              # prune.py:122-124: element int(0.9 N) of the descending sort
        kth, _ = lg_prune._select_mask(volume.detach().contiguous().float(), n - 1 - min(int(n * 0.9), n - 1), want_mask=False)
    v_list = torch.pow(volume / kth[0], v_pow)                                         # prune.py:126-127, the reference's own ops
    return v_list * imp_list


def _prune_points(self, mask):
    """GaussianModel.prune_points(mask), scene/gaussian_model.py:584-600."""
    from . import prune as lg_prune
    lg_prune.prune_points(self, mask)


def _prune_gaussians(self, percent, import_score):
    """GaussianModel.prune_gaussians(percent, import_score), scene/gaussian_model.py:776-782.  Float scores: threshold by one
    radix select.  Integer scores (prune_type "count" hands over the int32 hit counts, prune_finetune.py:229-232): the
    reference's own sort formulation (a float conversion could merge neighbouring counts), then the compaction."""
    import torch
    from . import prune as lg_prune
    if torch.is_tensor(import_score) and import_score.is_cuda and import_score.dtype == torch.float32:
        lg_prune.prune_gaussians(self, percent, import_score)
    else:
        lg_prune.prune_points(self, lg_prune.prune_mask(percent, import_score))
    from . import dp
    dp.assert_same_count(self.get_xyz.shape[0], "Gaussians after prune_gaussians", values=self.get_xyz)


class _LazyNegDist:
    """What `-torch.cdist(x, c, p=2)` evaluates to inside the patched vectree.vq module: the two operands, nothing computed.
    argmax over the last dimension -- the only thing the reference does with it at temperature 0 (vq.py:266 via gumbel_sample,
    :137 torch.argmax) -- is the nearest-code search on the matrix cores; anything else materialises the real tensor."""

    def __init__(self, a, b, negated=False):
        self.a, self.b, self.negated = a, b, negated

    def __neg__(self):
        return _LazyNegDist(self.a, self.b, not self.negated)

    def materialize(self):
        import torch
        d = torch.cdist(self.a, self.b, p=2)
        return -d if self.negated else d

    def nearest(self):
        from . import vq as lg_vq
        return lg_vq.nearest_code(self.a, self.b)

    def __getattr__(self, name):            # any other use: behave like the tensor the reference would have had
        return getattr(self.materialize(), name)


def _patch_vq(vq_mod):
    import torch

    class _TorchProxy:
        """Stands in for the name `torch` inside vectree/vq.py: cdist of two HIP float tensors (p = 2) is deferred, argmax of a
        deferred negated distance is the fused search; every other attribute is torch's own."""

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def cdist(a, b, p=2, *args, **kw):
            ok = (p == 2 and not args and not kw and torch.is_tensor(a) and torch.is_tensor(b) and a.is_cuda and b.is_cuda
                  and a.dim() in (2, 3) and a.dim() == b.dim() and 1 <= a.shape[-1] <= 63 and a.dtype == torch.float32 and b.dtype == torch.float32)
            return _LazyNegDist(a, b) if ok else torch.cdist(a, b, p, *args, **kw)

        @staticmethod
        def argmax(t, *args, **kw):
            dim = kw.get("dim", args[0] if args else None)
            if isinstance(t, _LazyNegDist):
                if t.negated and dim == -1 and not kw.get("keepdim", False):
                    return t.nearest()
                t = t.materialize()
            return torch.argmax(t, *args, **kw)

    orig_gumbel = vq_mod.gumbel_sample

    def gumbel_sample(t, temperature=1.0, dim=-1):
        """vectree/vq.py gumbel_sample: temperature 0 is a plain argmax (the only setting the reference's VecTree uses)."""
        if isinstance(t, _LazyNegDist):
            if temperature == 0 and dim == -1 and t.negated:
                return t.nearest()
            t = t.materialize()
        return orig_gumbel(t, temperature=temperature, dim=dim)

    _set(vq_mod, "torch", _TorchProxy(), "vectree.vq.torch (cdist -> deferred, argmax -> nearest_code)")
    _PATCHED.append((vq_mod, "gumbel_sample", orig_gumbel))
    vq_mod.gumbel_sample = gumbel_sample
    _REPORT["vectree.vq.gumbel_sample"] = {"old": "vectree.vq.gumbel_sample", "new": "lightgaussian_amd.run gumbel_sample -> lightgaussian_amd.vq.nearest_code"}


def patch_reference(verbose=False, data_parallel=False):
    """data_parallel=True (run.py --distributed): additionally hang lightgaussian_amd.dp on Scene / GaussianModel and record the
    visibility of every render() for the gradient exchange in front of optimizer.step().
    Import the reference's modules (they must be importable: run from / put on sys.path the reference checkout, with this
    repo on the path for the `diff_gaussian_rasterization` / `simple_knn` shims) and rebind the symbols listed in the module
    docstring.  Idempotent.  Returns a report {symbol: {"old": ..., "new": ..., "also_rebound_in": n}} (or {"skipped": why})."""
    from . import gaussian_renderer as lg_gr
    from . import loss_utils as lg_loss
    gr = _module("gaussian_renderer")
    if gr is not None:
        render = lg_gr.render
        if data_parallel:
            from . import dp
            render = _DP_RENDER.setdefault("fn", dp.wrap_render(lg_gr.render))
        _set(gr, "render", render, "gaussian_renderer.render")
        _set(gr, "count_render", lg_gr.count_render, "gaussian_renderer.count_render")
    lu = _module("utils.loss_utils")
    if lu is not None:
        _set(lu, "l1_loss", lg_loss.l1_loss, "utils.loss_utils.l1_loss")
        _set(lu, "ssim", lg_loss.ssim, "utils.loss_utils.ssim")
    pr = _module("prune")
    if pr is not None and os.path.abspath(getattr(pr, "__file__", "")).startswith(_ROOT + os.sep + "lightgaussian_amd"):
        pr = None                                                   # (our own module under that name: nothing to patch)
    if pr is not None:
        _set(pr, "prune_list", _prune_list, "prune.prune_list")
        _set(pr, "calculate_v_imp_score", _calculate_v_imp_score, "prune.calculate_v_imp_score")
    gm = _module("scene.gaussian_model")
    if gm is not None and hasattr(gm, "GaussianModel"):
        _set(gm.GaussianModel, "prune_points", _prune_points, "scene.gaussian_model.GaussianModel.prune_points")
        _set(gm.GaussianModel, "prune_gaussians", _prune_gaussians, "scene.gaussian_model.GaussianModel.prune_gaussians")
    if data_parallel:
        from . import dp
        sc = _module("scene")
        gm_cls = (getattr(gm, "GaussianModel", None) if gm is not None else None) or (getattr(sc, "GaussianModel", None) if sc is not None else None)
        dp.install(gm_cls, getattr(sc, "Scene", None) if sc is not None else None)
        _REPORT["data_parallel"] = {"old": "one trajectory per process", "new": "lightgaussian_amd.dp: camera shard per rank + gradient all-reduce before optimizer.step"}
    vq_mod = _module("vectree.vq")
    if vq_mod is not None and hasattr(vq_mod, "EuclideanCodebook") and not isinstance(getattr(vq_mod, "torch", None), type(None)) \
            and type(getattr(vq_mod, "torch")).__name__ != "_TorchProxy":
        _patch_vq(vq_mod)
    if verbose:
        for k, v in _REPORT.items():
            print(f"[lightgaussian_amd.run] {k}: {v}", file=sys.stderr)
    return dict(_REPORT)


def unpatch_reference():
    """Undo patch_reference() (reverse order)."""
    from . import dp
    dp.uninstall()
    _DP_RENDER.clear()
    while _PATCHED:
        owner, name, old = _PATCHED.pop()
        setattr(owner, name, old)
    _REPORT.clear()


_ADAM_INIT = {}


def fused_adam(enable=True):
    """run.py --fused-adam (opt-in; the optimizer is NOT part of the path this package replaces -- SURVEY 2, row 6 -- but it is what an
    iteration of the unmodified trainers spends most of its time in once the render is fast): the reference builds
    torch.optim.Adam(l, lr=0.0, eps=1e-15) (scene/gaussian_model.py:training_setup), which runs as ~8 multi-tensor kernels over the
    six parameter tensors and their moments; the same constructor with fused=True is ONE kernel per step.  Same update rule, float
    rounding aside; the optimizer-state surgery of prune / densify (exp_avg, exp_avg_sq by key) works on it unchanged.  Only Adam
    instances created while this is active, over CUDA parameters, and without an explicit fused / foreach argument are affected."""
    import torch
    if enable and "orig" not in _ADAM_INIT:
        orig = torch.optim.Adam.__init__
        _ADAM_INIT["orig"] = orig

        def __init__(self, params, *args, **kw):
            params = list(params)
            if "fused" not in kw and "foreach" not in kw:
                flat = [p for g in params for p in (g["params"] if isinstance(g, dict) else [g])]
                flat = [p for g in flat for p in (g if isinstance(g, (list, tuple)) else [g])]
                if flat and all(torch.is_tensor(p) and p.is_cuda and p.is_floating_point() for p in flat):
                    kw["fused"] = True
            orig(self, params, *args, **kw)

        torch.optim.Adam.__init__ = __init__
    elif not enable and "orig" in _ADAM_INIT:
        torch.optim.Adam.__init__ = _ADAM_INIT.pop("orig")


_EVENT_ELAPSED = {}


def event_timing(mode):
    """What `iter_start.elapsed_time(iter_end)` does when iter_end has not completed yet.  All three trainers evaluate it every
    iteration as an argument of training_report (prune_finetune.py:206, distill_train.py:161, train_densify_prune.py:162), right behind
    the loss.item() of their running average -- which, eagerly, has drained the device by then.  With --lazy-loss item() no longer
    waits for the backward, and torch raises "Both events must be completed before calculating elapsed time".
      "wait"  the call waits for its end event first: the trainers' timing stays what it was, the iteration is drained one line later
              than before (what --lazy-loss then saves is the ten tiny launches of the loss arithmetic)
      "skip"  (--no-iter-timing) an unfinished pair returns NaN instead of waiting: the host runs ahead of the device like a loop
              without the logging would (the tensorboard scalar `iter_time` becomes NaN); with it a trainer iteration of the bench
              (render + L1/D-SSIM + backward, loss.item() per iteration) goes from 1.77-1.85 ms to 1.54 ms
      None    torch's own method again"""
    import torch
    if mode is None:
        if "orig" in _EVENT_ELAPSED:
            torch.cuda.Event.elapsed_time = _EVENT_ELAPSED.pop("orig")
        _EVENT_ELAPSED.pop("mode", None)
        return
    if mode not in ("wait", "skip"):
        raise ValueError("event_timing: mode is 'wait', 'skip' or None")
    _EVENT_ELAPSED["mode"] = mode
    if "orig" not in _EVENT_ELAPSED:
        orig = torch.cuda.Event.elapsed_time
        _EVENT_ELAPSED["orig"] = orig

        def elapsed_time(self, end_event):
            if not end_event.query():
                if _EVENT_ELAPSED.get("mode") == "skip":
                    return float("nan")
                end_event.synchronize()
            return orig(self, end_event)

        torch.cuda.Event.elapsed_time = elapsed_time


def _redirect_model_path(argv, rank):
    """Ranks other than 0 of a data-parallel run write their outputs (cfg_args, point clouds, checkpoints, imp_score.npz: the
    trainers write them unconditionally) under <model_path>/.rank<r> instead of on top of rank 0's files.  -m / --model_path is
    rewritten in the trainer's argument list; without one the reference picks a random ./output/<uuid> per process anyway
    (prune_finetune.py prepare_output_and_logger)."""
    if rank == 0:
        return argv
    out = list(argv)
    for k, a in enumerate(out):
        if a in ("-m", "--model_path") and k + 1 < len(out):
            out[k + 1] = os.path.join(out[k + 1], f".rank{rank}")
        elif a.startswith("--model_path="):
            out[k] = "--model_path=" + os.path.join(a.split("=", 1)[1], f".rank{rank}")
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    distributed = no_patch = verbose = adam = lazy = no_timing = False
    backend = "nccl"
    dp_overlap = False
    weight_policy = None
    while argv and argv[0].startswith("--") and not argv[0].endswith(".py"):
        flag = argv.pop(0)
        if flag == "--distributed":
            distributed = True
        elif flag == "--no-patch":
            no_patch = True
        elif flag == "--verbose":
            verbose = True
        elif flag == "--dp-overlap":              # with --distributed: non-SH gradients all-reduced in ranges behind K9 (dp.install(overlap=True))
            dp_overlap = True
        elif flag == "--fused-adam":
            adam = True
        elif flag == "--lazy-loss":
            lazy = True
        elif flag == "--no-iter-timing":
            lazy = no_timing = True
        elif flag.startswith("--weight-policy="):
            from . import rasterizer
            weight_policy = flag.split("=", 1)[1]
            rasterizer.weight_policy_id(weight_policy)          # (raises on an unknown name before anything is set up)
        elif flag.startswith("--backend="):       # gloo: CPU tests of the launcher with a stand-in trainer (the rasterizer has no CPU path)
            backend = flag.split("=", 1)[1]
        else:
            raise SystemExit(f"lightgaussian_amd.run: unknown option {flag} (options: --distributed --dp-overlap --no-patch --verbose --fused-adam --lazy-loss --no-iter-timing --weight-policy=NAME, then the script and ITS arguments)")
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    if not os.path.exists(script):
        raise SystemExit(f"lightgaussian_amd.run: {script} does not exist")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # what `python script.py` sets up: the script's directory first; then this repo, so that the shims resolve
    sys.argv = [script] + (_redirect_model_path(argv[1:], rank) if distributed and world > 1 else argv[1:])
    for p in (_ROOT, os.path.dirname(script)):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, _ROOT)
    sys.path.insert(0, os.path.dirname(script))
    if distributed:
        # one process per GPU (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE).  The trainers' safe_state() forces
        # "cuda:0" (utils/general_utils.py:151): this rank's GPU is made the only visible one BEFORE the HIP runtime starts, so
        # that "cuda:0" is the right device on every rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if backend == "nccl" and "HIP_VISIBLE_DEVICES" not in os.environ and "CUDA_VISIBLE_DEVICES" not in os.environ and "ROCR_VISIBLE_DEVICES" not in os.environ:
            os.environ["HIP_VISIBLE_DEVICES"] = str(local)
            local = 0
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local)
        # (under a launcher -- RANK in the environment -- the group is created at world size 1 as well: harmless, every dp hook
        #  stays passive there unless dp.configure(force=True) sends the exchange through the collectives: the RCCL path on a 1-GPU box)
        if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
        # the launcher's ONE read of the LG_DP_* variables (dp.config_from_env); the step's path never looks at the environment
        from . import dp
        dp.config_from_env()
        if dp_overlap:
            dp.configure(overlap=True)
    if weight_policy is not None:
        from . import rasterizer
        rasterizer.set_option("weight_policy", rasterizer.weight_policy_id(weight_policy))     # process default of every count_render of the run
    if adam:
        fused_adam(True)
    if lazy:
        # l1_loss() / ssim() hand out lazy scalars (loss_utils.LazyLoss): the trainers' loss line costs no kernels and their
        # per-iteration loss.item() does not wait for the backward.  This (main) thread only; the trainers are single-threaded.
        from . import loss_utils as _lu
        _lu.set_lazy(True)
        event_timing("skip" if no_timing else "wait")
    if not no_patch:
        # --distributed is DATA-PARALLEL training (lightgaussian_amd.dp): a camera shard per rank, the gradients averaged over the
        # ranks in front of every optimizer.step(), prune_list sharded by camera; the ranks stay bit-identical replicas of one model
        report = patch_reference(verbose=verbose, data_parallel=distributed)
        missing = [k for k, v in report.items() if "skipped" in v]
        if missing and verbose:
            print(f"[lightgaussian_amd.run] not patched (module not importable): {missing}", file=sys.stderr)
    # the start-up heap (torch, numpy, the patched modules: several 10^5 objects) goes into CPython's permanent generation, so that a full
    # cyclic collection inside the training loop no longer walks it: ~35 ms of host stall per occurrence, with iterations of 1.4 ms
    # (bench.py --step-trace).  Nothing the trainer allocates afterwards is affected.
    import gc
    gc.collect()
    gc.freeze()
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        gc.unfreeze()
        if lazy:
            event_timing(None)
        if adam:
            fused_adam(False)
        if distributed:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dist.destroy_process_group()


if __name__ == "__main__":
    main()
