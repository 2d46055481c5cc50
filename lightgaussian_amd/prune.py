"""Global-Significance accumulation and prune mask -- host side.

Single-process functions mirror the reference one-to-one:
  prune_list(gaussians, scene, pipe, background)         <- prune.py:133-157
  calculate_v_imp_score(gaussians, imp_list, v_pow)      <- prune.py:112-128
  prune_mask(percent, import_score)                      <- scene/gaussian_model.py:776-782 (mask only; the
                                                            optimizer surgery of prune_points stays in GaussianModel)

prune_list_sharded() is the multi-GPU form: one process per GPU, Gaussians
replicated, the camera list split into contiguous blocks, and
  * gaussians_count : integer all-reduce(SUM) over RCCL -- exact for any GPU count;
  * important_score : mode="ordered" (default) moves per-view scores with one all_to_all so that rank j
                      owns a slice of the Gaussians for ALL views and adds them in the reference's view
                      order (prune.py:144-155: pop() from the end, in-place +=), then all_gathers.  The
                      result is bit-identical to the single-process loop for every world size, which is
                      what makes prune masks bit-identical across 1/2/4/8 GPUs.
                      mode="allreduce" does local sums + one fp32 all-reduce (fastest; deterministic for a
                      fixed world size, but float addition order differs from the reference loop).
"""
import torch
import torch.distributed as dist

from .gaussian_renderer import count_render


def calculate_v_imp_score(gaussians, imp_list, v_pow):
    """prune.py:112-128 verbatim semantics (index int(N*0.9) of the DESCENDING sort, no clamp)."""
    volume = torch.prod(gaussians.get_scaling, dim=1)
    index = int(len(volume) * 0.9)
    sorted_volume, _ = torch.sort(volume, descending=True)
    kth_percent_largest = sorted_volume[index]
    v_list = torch.pow(volume / kth_percent_largest, v_pow)
    v_list = v_list * imp_list
    return v_list


def prune_mask(percent, import_score):
    """Mask of GaussianModel.prune_gaussians (scene/gaussian_model.py:776-782): everything <= the
    value at index int(percent*(N-1)) of the ascending sort is pruned (ties included)."""
    sorted_tensor, _ = torch.sort(import_score, dim=0)
    index_nth_percentile = int(percent * (sorted_tensor.shape[0] - 1))
    value_nth_percentile = sorted_tensor[index_nth_percentile]
    return (import_score <= value_nth_percentile).squeeze()


def calculate_v_imp_score_select(gaussians, imp_list, v_pow):
    """calculate_v_imp_score with a radix SELECT instead of a full descending sort: the
    element at index int(N*0.9) of the descending order is the (N - index)-th smallest.  Same value, same v_list."""
    volume = torch.prod(gaussians.get_scaling, dim=1)
    n = volume.shape[0]
    index = int(n * 0.9)
    kth_percent_largest = torch.kthvalue(volume, n - index).values
    return torch.pow(volume / kth_percent_largest, v_pow) * imp_list


def prune_mask_select(percent, import_score):
    """prune_mask with one radix select: threshold = (index+1)-th smallest value, index = int(percent*(N-1))."""
    score = import_score.reshape(-1)
    index_nth_percentile = int(percent * (score.shape[0] - 1))
    value_nth_percentile = torch.kthvalue(score, index_nth_percentile + 1).values
    return (import_score <= value_nth_percentile).squeeze()


def prune_epilogue(gaussians, imp_list, v_pow, percent):
    """calculate_v_imp_score (prune.py:112-128) + the mask of prune_gaussians (scene/gaussian_model.py:776-782) in one
    device-resident pass of the HIP library (lg_prune_epilogue: two radix selects, no sort, no host read-back).
    Returns (v_list [N] float32, mask [N] bool, thresholds [2] = {kth volume, score threshold}); same values as
    calculate_v_imp_score(...) / prune_mask(percent, v_list).  CUDA/HIP tensors only -- no fallback."""
    import ctypes as C
    from . import _lib
    from . import rasterizer
    with torch.no_grad():
        scaling = gaussians.get_scaling.detach().contiguous().float()
        imp = imp_list.detach().reshape(-1).contiguous().float()
    if not (scaling.is_cuda and imp.is_cuda):
        raise RuntimeError("prune_epilogue runs on the MI355X HIP library only (no CPU fallback)")
    N = scaling.shape[0]
    if imp.shape[0] != N:
        raise ValueError(f"imp_list has {imp.shape[0]} entries for {N} Gaussians")
    lib = _lib.load()
    dev = scaling.device
    v_list = torch.empty(N, dtype=torch.float32, device=dev)
    mask = torch.empty(N, dtype=torch.uint8, device=dev)
    thresholds = torch.empty(2, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.lg_prune_scratch_bytes(N), dtype=torch.uint8, device=dev)
    flags = _lib.FLAG_PROFILE if rasterizer._OPTIONS["profile"] else 0
    _lib.check(lib.lg_prune_epilogue(N, scaling.data_ptr(), imp.data_ptr(), float(v_pow), float(percent), v_list.data_ptr(),
                                     mask.data_ptr(), thresholds.data_ptr(), scratch.data_ptr(), flags,
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return v_list, mask.bool(), thresholds


class _FrozenGetters:
    """The Gaussians do not change during a significance pass, so the activations and the cat() of
    GaussianModel's getters (scene/gaussian_model.py:98-118) are evaluated ONCE instead of once per view
    (the reference re-evaluates them in every count_render call: 1.15 GB of cat traffic per view at 3M).
    Same tensors, same values -- only the redundant recomputation is gone."""

    def __init__(self, pc):
        with torch.no_grad():
            self.get_xyz = pc.get_xyz.detach()
            self.get_opacity = pc.get_opacity.detach()
            self.get_scaling = pc.get_scaling.detach()
            self.get_rotation = pc.get_rotation.detach()
            self.get_features = pc.get_features.detach()
        self.active_sh_degree = pc.active_sh_degree
        self.max_sh_degree = pc.max_sh_degree
        self._pc = pc

    def get_covariance(self, scaling_modifier=1):
        return self._pc.get_covariance(scaling_modifier)


def _train_cameras(scene_or_list):
    if hasattr(scene_or_list, "getTrainCameras"):
        return scene_or_list.getTrainCameras().copy()
    return list(scene_or_list)


def prune_list(gaussians, scene, pipe, background, count_fn=count_render):
    """prune.py:133-157: sum of per-view (count, score) over all train cameras, popped from the END
    of the list; the first view's tensors are the accumulators."""
    viewpoint_stack = _train_cameras(scene)
    viewpoint_cam = viewpoint_stack.pop()
    render_pkg = count_fn(viewpoint_cam, gaussians, pipe, background)
    gaussian_list, imp_list = render_pkg["gaussians_count"], render_pkg["important_score"]
    for _ in range(len(viewpoint_stack)):
        viewpoint_cam = viewpoint_stack.pop()
        render_pkg = count_fn(viewpoint_cam, gaussians, pipe, background)
        gaussians_count, important_score = render_pkg["gaussians_count"].detach(), render_pkg["important_score"].detach()
        gaussian_list += gaussians_count
        imp_list += important_score
    return gaussian_list, imp_list


def shard_bounds(num_views, world_size, rank):
    """Contiguous block of the view SEQUENCE (sequence index s <-> cameras[V-1-s]) owned by `rank`."""
    return (num_views * rank) // world_size, (num_views * (rank + 1)) // world_size


def _count_views(seq, lo, hi, gaussians, pipe, background, count_fn, N, ncols, streams):
    """count_fn over seq[lo:hi].  Returns (count_sum int32 [N], per_view fp32 [hi-lo, ncols]) with per_view[k, :N] the
    score vector of view lo+k.  With streams > 1 the views are rendered by that many host threads, each on its own HIP
    stream (ctypes releases the GIL; the library is re-entrant per stream): the VALU-bound blend of one view overlaps the
    memory-latency-bound projection and the radix sort of another.  Results do not depend on the schedule: counts are
    integers and every view's scores land in their own row, to be summed in the reference's order afterwards."""
    dev = gaussians.get_xyz.device
    nv = max(hi - lo, 0)
    per_view = torch.zeros((nv, ncols), dtype=torch.float32, device=dev)
    streams = max(1, min(int(streams), nv)) if nv else 1
    if streams == 1:
        count_sum = torch.zeros(N, dtype=torch.int32, device=dev)
        with torch.no_grad():
            for k in range(nv):
                pkg = count_fn(seq[lo + k], gaussians, pipe, background)
                count_sum += pkg["gaussians_count"].detach().to(torch.int32)
                per_view[k, :N] = pkg["important_score"].detach()
        return count_sum, per_view
    import threading
    main = torch.cuda.current_stream(dev)
    pool = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    partial = [torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(streams)]
    errors = []

    def work(w):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(pool[w]), torch.no_grad():
                pool[w].wait_stream(main)                 # the frozen getters / zeroed buffers were produced on `main`
                for k in range(w, nv, streams):
                    pkg = count_fn(seq[lo + k], gaussians, pipe, background)
                    partial[w] += pkg["gaussians_count"].to(torch.int32)
                    per_view[k, :N] = pkg["important_score"]
        except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(w,)) for w in range(streams)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    for st in pool:
        main.wait_stream(st)
    count_sum = partial[0]
    for p in partial[1:]:
        count_sum += p
    return count_sum, per_view


def _ordered_sum(rows):
    """acc = rows[0]; acc += rows[1]; ... : the reference's sequential in-place float adds (prune.py:144-155).
    On the GPU one lg_ordered_sum launch (same additions, same order); CPU tensors (gloo tests) take the literal loop."""
    if rows.is_cuda and rows.dtype == torch.float32 and rows.stride(1) == 1:
        import ctypes as C
        from . import _lib
        out = torch.empty(rows.shape[1], dtype=torch.float32, device=rows.device)
        _lib.check(_lib.load().lg_ordered_sum(rows.shape[0], rows.shape[1], rows.data_ptr(), rows.stride(0), out.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream(rows.device).cuda_stream)))
        return out
    acc = rows[0].clone()
    for s in range(1, rows.shape[0]):
        acc += rows[s]
    return acc


def prune_list_sharded(gaussians, scene, pipe, background, group=None, mode="ordered", count_fn=count_render, force_collectives=False,
                       streams=3):
    """Camera-sharded prune_list.  Every rank passes the SAME full camera list and the same
    Gaussians; returns the same (gaussian_list, imp_list) on every rank.  Without an initialised process
    group (or at world size 1, unless force_collectives) it is the single-process loop with frozen getters.
    streams: views in flight per rank (host threads x HIP streams, _count_views); 1 = the plain sequential loop.
    Measured at C3 on one MI355X: 1 -> 1158, 2 -> 1061, 3 -> 1341, 4 -> 1326 views/s (identical results)."""
    from . import rasterizer
    prev = rasterizer._OPTIONS["skip_color_in_count"]
    rasterizer.set_option("skip_color_in_count", True)   # the pass discards the images: do not read 192 B of SH per Gaussian per view
    try:
        return _prune_list_sharded(gaussians, scene, pipe, background, group, mode, count_fn, force_collectives, streams)
    finally:
        rasterizer.set_option("skip_color_in_count", prev)


def _prune_list_sharded(gaussians, scene, pipe, background, group, mode, count_fn, force_collectives, streams):
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    if mode not in ("ordered", "allreduce"):
        raise ValueError(f"unknown mode {mode!r}")
    gaussians = _FrozenGetters(gaussians)
    cams = _train_cameras(scene)
    V = len(cams)
    seq = cams[::-1]  # sequence order of the reference loop (pop() from the end)
    N = gaussians.get_xyz.shape[0]
    dev = gaussians.get_xyz.device
    if world == 1 and not (distributed and force_collectives):
        if not gaussians.get_xyz.is_cuda or streams <= 1 or V < 2:
            return prune_list(gaussians, scene, pipe, background, count_fn)
        count_sum, per_view = _count_views(seq, 0, V, gaussians, pipe, background, count_fn, N, N, streams)
        # prune.py:136-141: the first view's own tensors are the accumulators (count dtype = what the rasterizer returns)
        return count_sum, _ordered_sum(per_view)

    lo, hi = shard_bounds(V, world, rank)
    chunk = (N + world - 1) // world
    count_sum, per_view = _count_views(seq, lo, hi, gaussians, pipe, background, count_fn, N, world * chunk,
                                       streams if gaussians.get_xyz.is_cuda else 1)
    dist.all_reduce(count_sum, op=dist.ReduceOp.SUM, group=group)

    if mode == "allreduce":
        local_score = _ordered_sum(per_view[:, :N]) if hi > lo else torch.zeros(N, dtype=torch.float32, device=dev)
        dist.all_reduce(local_score, op=dist.ReduceOp.SUM, group=group)
        return count_sum, local_score

    # ordered: rank j receives Gaussian slice j of every view, in sequence order
    v_local = hi - lo
    send = per_view.view(v_local, world, chunk).permute(1, 0, 2).contiguous().view(world * v_local, chunk)
    sizes_out = [shard_bounds(V, world, r)[1] - shard_bounds(V, world, r)[0] for r in range(world)]
    recv = torch.empty((V, chunk), dtype=torch.float32, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=sizes_out, input_split_sizes=[v_local] * world, group=group)
    acc = _ordered_sum(recv)
    gathered = torch.empty(world * chunk, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(gathered, acc, group=group)
    return count_sum, gathered[:N].contiguous()
