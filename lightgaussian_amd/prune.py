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


def prune_list_sharded(gaussians, scene, pipe, background, group=None, mode="ordered", count_fn=count_render, force_collectives=False):
    """Camera-sharded prune_list.  Every rank passes the SAME full camera list and the same
    Gaussians; returns the same (gaussian_list, imp_list) on every rank.  Without an initialised process
    group (or at world size 1, unless force_collectives) it is the single-process loop with frozen getters."""
    from . import rasterizer
    prev = rasterizer._OPTIONS["skip_color_in_count"]
    rasterizer.set_option("skip_color_in_count", True)   # the pass discards the images: do not read 192 B of SH per Gaussian per view
    try:
        return _prune_list_sharded(gaussians, scene, pipe, background, group, mode, count_fn, force_collectives)
    finally:
        rasterizer.set_option("skip_color_in_count", prev)


def _prune_list_sharded(gaussians, scene, pipe, background, group, mode, count_fn, force_collectives):
    if not (dist.is_available() and dist.is_initialized()):
        return prune_list(_FrozenGetters(gaussians), scene, pipe, background, count_fn)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not force_collectives:
        return prune_list(_FrozenGetters(gaussians), scene, pipe, background, count_fn)
    if mode not in ("ordered", "allreduce"):
        raise ValueError(f"unknown mode {mode!r}")
    gaussians = _FrozenGetters(gaussians)
    cams = _train_cameras(scene)
    V = len(cams)
    seq = cams[::-1]  # sequence order of the reference loop (pop() from the end)
    lo, hi = shard_bounds(V, world, rank)
    N = gaussians.get_xyz.shape[0]
    dev = gaussians.get_xyz.device

    count_sum = torch.zeros(N, dtype=torch.int32, device=dev)
    chunk = (N + world - 1) // world
    if mode == "ordered":
        per_view = torch.zeros((max(hi - lo, 0), world * chunk), dtype=torch.float32, device=dev)
    else:
        local_score = None
    with torch.no_grad():
        for k, s in enumerate(range(lo, hi)):
            pkg = count_fn(seq[s], gaussians, pipe, background)
            count_sum += pkg["gaussians_count"].detach().to(torch.int32)
            sc = pkg["important_score"].detach()
            if mode == "ordered":
                per_view[k, :N] = sc
            else:
                local_score = sc.clone() if local_score is None else local_score.add_(sc)

    dist.all_reduce(count_sum, op=dist.ReduceOp.SUM, group=group)

    if mode == "allreduce":
        if local_score is None:
            local_score = torch.zeros(N, dtype=torch.float32, device=dev)
        dist.all_reduce(local_score, op=dist.ReduceOp.SUM, group=group)
        return count_sum, local_score

    # ordered: rank j receives Gaussian slice j of every view, in sequence order
    v_local = hi - lo
    send = per_view.view(v_local, world, chunk).permute(1, 0, 2).contiguous().view(world * v_local, chunk)
    sizes_out = [shard_bounds(V, world, r)[1] - shard_bounds(V, world, r)[0] for r in range(world)]
    recv = torch.empty((V, chunk), dtype=torch.float32, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=sizes_out, input_split_sizes=[v_local] * world, group=group)
    acc = recv[0].clone()
    for s in range(1, V):  # the reference's sequential in-place float adds
        acc += recv[s]
    gathered = torch.empty(world * chunk, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(gathered, acc, group=group)
    return count_sum, gathered[:N].contiguous()
