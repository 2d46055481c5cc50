"""A whole training step as ONE HIP graph.

With the sync-free forward (lg_forward_bounded, nothing read back) a step -- render() with the getters inside K1/K9, the
photometric loss, the whole backward -- is a fixed sequence of stream-ordered kernel launches with fixed grids.  GraphedStep
captures it once per (model size, image shape, field of view) with torch.cuda.graph and replays it for other cameras by
overwriting static camera / target tensors in place: ~25 launches per step become one hipGraphLaunch.  At the benchmark size
(3 M Gaussians, 1080p) the device is busy end to end either way; for small scenes (finetune / distill at a few 100 k
Gaussians, config C5) the step is launch-bound and the replay is what removes that bound.

The captured step matches prune_finetune.py:150-170's per-iteration body between `render()` and `loss.backward()`; the
optimizer step stays outside (eager), after the caller has had a chance to look at overflowed():

    step = GraphedStep(gaussians, pipe, background, loss="l1_dssim", lambda_dssim=opt.lambda_dssim)
    for cam in cameras:
        loss = step(cam, cam.original_image)        # gradients are in gaussians._xyz.grad, ...
        optimizer.step()

A view that does not fit the binning capacity the graph was captured with is abandoned on the device (image = background,
gradients zero; lg_forward_bounded).  The graph's K2 writes the four status words of the view straight into PINNED HOST
memory; after enqueueing the replay __call__ polls word 0 -- it arrives as soon as K2 has run, while the sort and the blend
kernels of the same replay are still executing, so the check costs no device idle time (the graph form of the
"validated" forward).  An abandoned step is re-run eagerly through the exact path before __call__ returns, and the next call
re-captures with the larger capacity.  There is no CPU fallback: tensors must live on a HIP device.
"""
import torch

from . import loss_utils
from . import rasterizer as _rast
from .gaussian_renderer import render

_PARAMS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


class _StaticCam:
    """The camera fields render() reads (gaussian_renderer/__init__.py:48-60), with matrices that are overwritten in place."""

    def __init__(self, cam, dev):
        self.image_width, self.image_height = int(cam.image_width), int(cam.image_height)
        self.FoVx, self.FoVy = float(cam.FoVx), float(cam.FoVy)
        self.world_view_transform = cam.world_view_transform.detach().to(dev, torch.float32).clone()
        self.full_proj_transform = cam.full_proj_transform.detach().to(dev, torch.float32).clone()
        self.camera_center = cam.camera_center.detach().to(dev, torch.float32).clone()

    def key(self):
        return (self.image_width, self.image_height, self.FoVx, self.FoVy)

    def load(self, cam):
        self.world_view_transform.copy_(cam.world_view_transform, non_blocking=True)
        self.full_proj_transform.copy_(cam.full_proj_transform, non_blocking=True)
        self.camera_center.copy_(cam.camera_center, non_blocking=True)


class GraphedStep:
    def __init__(self, pc, pipe, bg_color, loss="l1", lambda_dssim=0.2, scaling_modifier=1.0, check="every"):
        """loss: "l1" (L1 only) or "l1_dssim" ((1 - lambda) L1 + lambda (1 - SSIM), prune_finetune.py:161-164).
        check: "every" (default) looks at the replay's status words after each step and repairs an overflow before
        returning; "never" leaves that to the caller (overflowed())."""
        if loss not in ("l1", "l1_dssim"):
            raise ValueError("loss must be 'l1' or 'l1_dssim'")
        self.pc, self.pipe, self.bg, self.mod = pc, pipe, bg_color, float(scaling_modifier)
        self.loss_kind, self.lam, self.check = loss, float(lambda_dssim), check
        self._graphs = {}          # camera key -> (graph, static camera, static target, static loss, status tensor, model id, capacity key, grads, leaf view)
        self.replays = self.captures = self.repairs = 0

    # -- the step itself (eager form; also what gets captured) -------------------------------------------------------
    def _body(self, cam, gt, model=None):
        model = self.pc if model is None else model
        for n in _PARAMS:
            getattr(model, n).grad = None
        image = render(cam, model, self.pipe, self.bg, self.mod)["render"]
        if self.loss_kind == "l1":
            loss = loss_utils.l1_loss_only(image, gt)
        else:
            loss = loss_utils.l1_dssim_loss(image, gt, self.lam)[0]
        loss.backward()
        return loss.detach()

    def _capture(self, cam, gt):
        dev = self.pc._xyz.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedStep needs the model on a HIP device (torch 'cuda'); there is no CPU path")
        static_cam, static_gt = _StaticCam(cam, dev), gt.detach().to(dev, torch.float32).clone()
        batch = _rast.PendingBatch()
        with _rast.options(sync_free="validated"):
            self._body(static_cam, static_gt)                     # learns the binning capacity of this shape (exact path first)
        with _rast.options(sync_free=True, pending=batch):
            # The step runs on fresh autograd LEAVES that share the parameters' storage (parallel._LeafView), one set for the
            # warm-up and one for the capture: an AccumulateGrad node remembers the stream it was created on, and the
            # parameters' own nodes -- alive for as long as the caller holds any tensor of an earlier eager step -- were
            # created on the caller's stream; backward inside the capture would then synchronise with that stream (with the
            # legacy default stream: an invalid capture; this crashed hipStreamEndCapture).  The views' nodes are born on the
            # stream that uses them.  Gradients are handed to the parameters' .grad after every replay.
            from .parallel import _LeafView
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                         # warm-up off the capture stream, as graph capture asks for
                self._body(static_cam, static_gt, _LeafView(self.pc))
            torch.cuda.current_stream(dev).wait_stream(side)
            batch.resolve()
            view = _LeafView(self.pc)                             # its .grad tensors are allocated inside the graph's pool
            status = torch.full((4,), -1, dtype=torch.int32).pin_memory()   # written by K2 of every replay, polled by the host
            graph = torch.cuda.CUDAGraph()
            with _rast.options(status_override=status):
                with torch.cuda.graph(graph):
                    static_loss = self._body(static_cam, static_gt, view)
            with batch.lock:
                pend, batch.items = batch.items, []
            bounded = [p for p in pend if p[0] is status]
            if len(bounded) != 1:
                raise RuntimeError("GraphedStep: the captured step did not go through the capacity-bounded forward "
                                   "(prefiltered rasterizer settings or the pair key format have no bounded form)")
        self.captures += 1
        grads = [getattr(view, n).grad for n in _PARAMS]          # static: every replay writes these tensors
        for n, t in zip(_PARAMS, grads):
            getattr(self.pc, n).grad = t
        entry = (graph, static_cam, static_gt, static_loss, bounded[0][0], self._model_id(), bounded[0][1], grads, view)
        self._graphs[static_cam.key()] = entry
        return entry

    def __call__(self, cam, gt):
        """One step for camera `cam` against target image `gt` [3, H, W].  Returns the loss (a 0-d tensor that the next call
        overwrites); the gradients are in the parameters' .grad."""
        key = (int(cam.image_width), int(cam.image_height), float(cam.FoVx), float(cam.FoVy))
        entry = self._graphs.get(key)
        if entry is not None and entry[5] != self._model_id():
            entry = None                                          # the model was pruned / densified / reloaded: capture again
        if entry is None:
            dkey = (self.pc._xyz.device.index, int(self.pc._xyz.shape[0]), key[0], key[1])
            if _rast._CAPACITY.get(dkey, 0) < 0:
                return self._eager(cam, gt)                       # a depth beyond max_depth was seen for this shape: exact path only
            entry = self._capture(cam, gt)
        graph, static_cam, static_gt, static_loss, status, _n, cap_key, grads, _view = entry
        static_cam.load(cam)
        static_gt.copy_(gt, non_blocking=True)
        words = status.numpy()                                    # the pinned words themselves (no copy)
        words[0] = -1                                             # "not arrived"; K2 of this replay overwrites it
        graph.replay()
        self.replays += 1
        for n, t in zip(_PARAMS, grads):                          # (a zero_grad(set_to_none=True) in between detached them)
            getattr(self.pc, n).grad = t
        if self.check == "every" and self.overflowed(key):
            return self._repair(cam, gt, key, cap_key, status)
        return static_loss

    def _eager(self, cam, gt):
        with _rast.options(sync_free="validated"):
            return self._body(cam, gt)

    def _model_id(self):
        """The graph holds the parameters' addresses: any new storage (prune_points, densification, load_ply) invalidates it."""
        return tuple((getattr(self.pc, n).data_ptr(), tuple(getattr(self.pc, n).shape)) for n in _PARAMS) + (int(self.pc.active_sh_degree),)

    def overflowed(self, key=None):
        """True when the last replay (of camera shape `key`, default: any) was abandoned on the device.  Waits for that
        replay's K2 only (its status words land in pinned host memory), not for the replay."""
        entries = [self._graphs[key]] if key is not None else list(self._graphs.values())
        bad = False
        for e in entries:
            words = e[4].numpy()
            spins = 0
            while int(words[0]) == -1:                            # K1 + K2 of the replay: a few 100 us at most
                spins += 1
                if spins > 200000:                                # (~1 s) something else holds the device: fall back to a full wait
                    torch.cuda.current_stream(self.pc._xyz.device).synchronize()
                    if int(words[0]) == -1:
                        raise RuntimeError("GraphedStep: the replay finished without writing its status words")
            bad = bad or int(words[0]) != 0
        return bad

    def _repair(self, cam, gt, key, cap_key, status):
        """The view did not fit: raise the capacity from the instance count the device reported (or pin the shape to the
        exact path when it was the depth bound), redo the step eagerly, and drop the graph so that the next call re-captures."""
        torch.cuda.current_stream(self.pc._xyz.device).synchronize()   # the abandoned replay still owns the static buffers
        flags, _v, _d, R = [int(x) & 0xFFFFFFFF for x in status.tolist()]
        with _rast._CAP_LOCK:
            _rast._CAPACITY[cap_key] = -1 if (flags & 2) else max(_rast._CAPACITY.get(cap_key, 0),
                                                                  int(R * _rast.resolve_options()["capacity_margin"]) + 4096)
        del self._graphs[key]
        self.repairs += 1
        return self._eager(cam, gt)
