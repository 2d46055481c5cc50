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


def _select_mask(values, rank, want_mask=True):
    """rank-th smallest element (0-based) of a 1-D float32 device tensor by the HIP radix select (lg_select_mask: four 8-bit
    histogram passes, no sort, no host read-back) and, optionally, mask = values <= that element.  Returns (value [1], mask)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = values.shape[0]
    dev = values.device
    out = torch.empty(1, dtype=torch.float32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev) if want_mask else None
    scratch = torch.empty(lib.lg_prune_scratch_bytes(n), dtype=torch.uint8, device=dev)
    _lib.check(lib.lg_select_mask(n, values.data_ptr(), int(rank), None if mask is None else mask.data_ptr(), out.data_ptr(),
                                  scratch.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out, mask


def prune_epilogue(gaussians, imp_list, v_pow, percent, fused_pow=False):
    """calculate_v_imp_score (prune.py:112-128) + the mask of prune_gaussians (scene/gaussian_model.py:776-782), device
    resident: both order statistics are HIP radix SELECTS (no sort, no host read-back of the thresholds).
    Returns (v_list [N] float32, mask [N] bool, thresholds [2] = {kth volume, score threshold}).

    fused_pow=False (default): the two arithmetic steps between the selects -- torch.prod(scaling, 1) and
        torch.pow(volume / kth, v_pow) * imp_list -- are evaluated by the SAME torch ops the reference runs, so v_list and the
        mask are bit-identical to calculate_v_imp_score(...) / prune_mask(percent, v_list) (Hamming distance 0, asserted in
        tests/test_gpu_prune_epilogue.py); a select returns exactly the element a sort puts at that index.
    fused_pow=True: one library call (lg_prune_epilogue) that also evaluates powf(volume / kth, v_pow) * imp in the kernel.
        Its v_list is within 1 ulp of torch.pow's (the device powf and torch's pow kernel round differently in the last
        bit), so elements exactly at the threshold can fall on the other side: NOT the reference's mask, bit for bit.
    CUDA/HIP tensors only -- no fallback."""
    import ctypes as C
    from . import _lib
    from . import rasterizer
    with torch.no_grad():
        scaling = gaussians.get_scaling.detach().contiguous().float()
        imp = imp_list.detach().reshape(-1).contiguous().float()
    if not (scaling.is_cuda and imp.is_cuda):
        raise RuntimeError("prune_epilogue runs on the MI355X HIP library only (no CPU fallback)")
    N = scaling.shape[0]
    if N <= 0:
        raise Exception("prune epilogue needs N >= 1 (the reference indexes an empty sort)")
    if imp.shape[0] != N:
        raise ValueError(f"imp_list has {imp.shape[0]} entries for {N} Gaussians")
    if not (0.0 <= float(percent) <= 1.0):
        raise Exception("prune_percent must be in [0, 1]")
    if not fused_pow:
        with torch.no_grad():
            volume = torch.prod(scaling, dim=1)                              # prune.py:120
            index = int(N * 0.9)                                             # prune.py:122: element `index` of the DESCENDING sort
            kth, _ = _select_mask(volume, N - 1 - min(index, N - 1), want_mask=False)
            v_list = torch.pow(volume / kth[0], v_pow) * imp                 # prune.py:126-127, the reference's own ops
            thr, mask = _select_mask(v_list, int(float(percent) * (N - 1)))  # scene/gaussian_model.py:778-781
        return v_list, mask.bool(), torch.cat((kth, thr))
    lib = _lib.load()
    dev = scaling.device
    v_list = torch.empty(N, dtype=torch.float32, device=dev)
    mask = torch.empty(N, dtype=torch.uint8, device=dev)
    thresholds = torch.empty(2, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.lg_prune_scratch_bytes(N), dtype=torch.uint8, device=dev)
    flags = _lib.FLAG_PROFILE if rasterizer.resolve_options()["profile"] else 0
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
    # (a data-parallel run shards Scene.getTrainCameras per rank, lightgaussian_amd.dp; the significance pass needs the whole list)
    if hasattr(scene_or_list, "_lg_all_train_cameras"):
        return scene_or_list._lg_all_train_cameras().copy()
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
    """Contiguous block of a view sequence owned by `rank` (plain partition helper; the significance pass itself uses the
    block-cyclic schedule of round_views so that its memory does not grow with the number of views)."""
    return (num_views * rank) // world_size, (num_views * (rank + 1)) // world_size


def num_rounds(num_views, world_size, block):
    return (num_views + block * world_size - 1) // (block * world_size)


def round_views(num_views, world_size, rank, block, t):
    """Sequence indices [lo, hi) rendered by `rank` in round t: round t covers the next block*world views of the reference's
    sequence, `block` consecutive ones per rank in rank order -- so the views of a round, concatenated over the ranks, are in
    sequence order and a running sum can absorb them round by round (memory O(block * N), not O(V * N))."""
    base = t * block * world_size + rank * block
    return min(base, num_views), min(base + block, num_views)


class _ViewRunner:
    """Renders lists of views with count_fn, `streams` views in flight on as many HIP streams, so that the VALU-bound blend
    of one view overlaps the memory-bound projection and the radix sort of another.  Default: ONE host thread issues view k
    onto stream k % streams through the capacity-bounded forward (rasterizer option sync_free: no read-back of the instance
    count, the thread runs ahead of the GPU); a view that outgrew its binning capacity contributes zeros on the device and is
    re-rendered on the exact path after the one status check per block.  host_threads=True is the round-1 scheme (a host
    thread per stream, each blocking in its own read-back).  Results do not depend on the schedule: counts are integers
    (per-stream partial sums, added at the end) and every view's score vector lands in its own row of the caller's buffer,
    to be summed in the reference's order afterwards."""

    def __init__(self, gaussians, pipe, background, count_fn, N, streams, host_threads=False, options=None):
        self.g, self.pipe, self.bg, self.count_fn, self.N = gaussians, pipe, background, count_fn, N
        self.options = dict(options or {})     # rasterizer options of every forward this runner issues (thread-local: rasterizer.options)
        self.dev = gaussians.get_xyz.device
        self.streams = max(1, int(streams)) if self.dev.type == "cuda" else 1
        self.host_threads = host_threads
        self.partial = [torch.zeros(N, dtype=torch.int32, device=self.dev) for _ in range(self.streams)]
        self.fused = count_fn is count_render and self.dev.type == "cuda"      # (test stand-ins for count_fn return their own tensors)
        self.pool = [torch.cuda.Stream(device=self.dev) for _ in range(self.streams)] if self.streams > 1 else []

    def _one(self, view, w, rows, k):
        if self.fused:
            # the forward writes the score straight into the caller's row and adds the hit count to this stream's running sum inside
            # its own score kernel (rasterizer options score_out / count_sum): no `+=` and no row copy launched per view
            from . import rasterizer
            with rasterizer.options(score_out=rows[k, :self.N], count_sum=self.partial[w]):
                self.count_fn(view, self.g, self.pipe, self.bg)
            return
        pkg = self.count_fn(view, self.g, self.pipe, self.bg)
        self.partial[w] += pkg["gaussians_count"].detach().to(torch.int32)
        rows[k, :self.N] = pkg["important_score"].detach()

    def run(self, views, rows):
        """rows[k, :N] = important_score of views[k]; gaussians_count accumulates into the partial sums."""
        nv = len(views)
        if nv == 0:
            return
        from . import rasterizer
        if self.streams == 1 or nv == 1:
            with torch.no_grad(), rasterizer.options(**self.options):
                for k in range(nv):
                    self._one(views[k], 0, rows, k)
            return
        main = torch.cuda.current_stream(self.dev)
        K = min(self.streams, nv)
        if self.host_threads:
            import threading
            errors = []

            def work(w):
                try:
                    torch.cuda.set_device(self.dev)
                    with torch.cuda.stream(self.pool[w]), torch.no_grad(), rasterizer.options(**self.options):
                        self.pool[w].wait_stream(main)    # frozen getters / the running sum were produced on `main`
                        for k in range(w, nv, K):
                            self._one(views[k], w, rows, k)
                except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread
                    errors.append(e)

            threads = [threading.Thread(target=work, args=(w,)) for w in range(K)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            for st in self.pool[:K]:
                main.wait_stream(st)
            return
        # one host thread, sync-free forwards: the options apply to THIS thread only and the status words of these views are
        # collected in a batch of their own, tagged with the view index -- whatever else the process renders meanwhile (another
        # thread in render(), a count_fn that issues several forwards or none) cannot be mistaken for one of them
        batch = rasterizer.PendingBatch()
        for st in self.pool[:K]:
            st.wait_stream(main)
        with torch.no_grad(), rasterizer.options(**dict(self.options, sync_free=True, pending=batch)):
            for k in range(nv):
                with torch.cuda.stream(self.pool[k % K]), rasterizer.options(tag=k):
                    self._one(views[k], k % K, rows, k)
        for st in self.pool[:K]:
            main.wait_stream(st)
        redo = sorted({tag for tag, bad in batch.resolve() if bad})     # ONE host sync per block of views
        if redo:
            # an abandoned view added zeros to the counts and wrote a zero score row: render it again, exact forward
            with torch.no_grad(), rasterizer.options(**dict(self.options, sync_free=False)):
                for k in redo:
                    self._one(views[k], 0, rows, k)

    def count_sum(self):
        total = self.partial[0]
        for p in self.partial[1:]:
            total += p
        return total


def _ordered_sum(rows, out=None):
    """out = rows[0]; out += rows[1]; ... : the reference's sequential in-place float adds (prune.py:144-155).  `out` may be
    rows[0] itself (running sum kept in row 0).  On the GPU one lg_ordered_sum launch (same additions, same order); CPU
    tensors (gloo tests) take the literal loop."""
    if out is None:
        out = torch.empty(rows.shape[1], dtype=rows.dtype, device=rows.device)
    if rows.is_cuda and rows.dtype == torch.float32 and rows.stride(1) == 1 and out.is_contiguous():
        import ctypes as C
        from . import _lib
        _lib.check(_lib.load().lg_ordered_sum(rows.shape[0], rows.shape[1], rows.data_ptr(), rows.stride(0), out.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream(rows.device).cuda_stream)))
        return out
    if out.data_ptr() != rows[0].data_ptr():
        out.copy_(rows[0])
    for s in range(1, rows.shape[0]):
        out += rows[s]
    return out


def prune_list_sharded(gaussians, scene, pipe, background, group=None, mode="ordered", count_fn=count_render, force_collectives=False,
                       streams=4, block=24, local_only=False, host_threads=False, stats=None, weight_policy=None):
    """Camera-sharded prune_list.  Every rank passes the SAME full camera list and the same Gaussians; returns the same
    (gaussian_list, imp_list) on every rank, bit-identical to the reference loop (mode="ordered") for every world size.
    Without an initialised process group (or at world size 1 unless force_collectives, or with local_only=True inside a
    distributed job) it is the single-process loop with frozen getters.
    streams: views in flight per rank (_ViewRunner: one host thread, sync-free forwards; host_threads=True = a thread per
             stream with exact forwards); 1 = the plain sequential loop.
    stats:   optional dict; receives "exchange_seconds" (time of this rank's collectives: device events on the current stream for HIP
             tensors, wall clock for CPU tensors), "collectives" and "world".
    weight_policy: per-hit weight of important_score for the forwards of THIS pass ("opacity" | "one" | "alpha" | "alpha_t" or a
             _lib.WEIGHT_* number; None = the rasterizer option in force, default "opacity").  Every policy gives per-view scores that
             are pure functions of the view (integer-derived or Q24.40 fixed-point sums, no float atomics), so the ordered exchange
             below makes scores and masks bit-identical across world sizes for all four.
    block:   views per rank per round.  The pass keeps a running sum and absorbs the views round by round in the reference's
             order, so scratch memory is O(block * N) floats per rank whatever the number of views (the reference's loop is
             O(N); the first version of this function held all V score vectors, O(V * N)).
    Measured at C3 on one MI355X (identical results): round 1, a host thread per stream: streams 1 -> 1158, 2 -> 1061, 3 -> 1341,
    4 -> 1326 views/s; round 2, one host thread: 3 -> 1540, 4 -> 1579, 6 -> 1513 (heavy-tailed scene: 1336 / 1412 / 1299)."""
    # the pass discards the images: its forwards do not read 192 B of SH per Gaussian per view (a per-call option of the
    # forwards THIS pass issues -- the process defaults are not touched, other threads render as before)
    opts = {"skip_color_in_count": True}
    if weight_policy is not None:
        from . import rasterizer
        opts["weight_policy"] = rasterizer.weight_policy_id(weight_policy)
    return _prune_list_sharded(gaussians, scene, pipe, background, group, mode, count_fn, force_collectives, streams, max(1, int(block)),
                               local_only, host_threads, opts, stats)


class _CollectiveTimer:
    """Brackets the collectives of one pass: hipEvents on the current stream (HIP tensors) or perf_counter (gloo tests)."""

    def __init__(self, dev, stats):
        self.dev, self.stats, self.pairs, self.wall, self.n = dev, stats, [], 0.0, 0

    def __call__(self, fn, *a, **kw):
        if self.stats is None:
            return fn(*a, **kw)
        self.n += 1
        if self.dev.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.pairs.append((e0, e1))
            return out
        import time
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        self.wall += time.perf_counter() - t0
        return out

    def finish(self, world):
        if self.stats is None:
            return
        if self.pairs:
            torch.cuda.synchronize(self.dev)
            self.wall += sum(a.elapsed_time(b) for a, b in self.pairs) * 1e-3
        self.stats.update({"exchange_seconds": self.wall, "collectives": self.n, "world": world})


def _prune_list_sharded(gaussians, scene, pipe, background, group, mode, count_fn, force_collectives, streams, block, local_only, host_threads,
                        options=None, stats=None):
    distributed = dist.is_available() and dist.is_initialized() and not local_only
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
    f32 = dict(dtype=torch.float32, device=dev)
    runner = _ViewRunner(gaussians, pipe, background, count_fn, N, streams, host_threads, options)
    timed = _CollectiveTimer(dev, stats)
    if world == 1 and not (distributed and force_collectives):
        timed.finish(1)
        if V == 0:
            raise IndexError("pop from empty list")     # what the reference's viewpoint_stack.pop() raises
        # rows[0] = running sum.  It starts at +0: 0 + s == s bit for bit (scores are sums of non-negative weights), so the
        # result equals the reference's "first view's tensor is the accumulator" (prune.py:136-141)
        rows = torch.zeros((min(block, V) + 1, N), **f32)
        for c0 in range(0, V, block):
            views = seq[c0:c0 + block]
            runner.run(views, rows[1:])
            _ordered_sum(rows[:1 + len(views)], out=rows[0])
        return runner.count_sum(), rows[0].clone()

    chunk = (N + world - 1) // world                   # Gaussian slice owned by each rank in the ordered exchange
    # rows[1:] = my views of the current round (columns >= N stay zero); rows[0] = running local sum (mode "allreduce")
    rows = torch.zeros((block + 1, world * chunk), **f32)
    if mode == "ordered":
        recv = torch.zeros((1 + block * world, chunk), **f32)   # row 0 = running sum of MY slice over all views so far
    for t in range(num_rounds(V, world, block)):
        spans = [round_views(V, world, r, block, t) for r in range(world)]
        sizes = [hi - lo for lo, hi in spans]
        lo, hi = spans[rank]
        mine = hi - lo
        runner.run(seq[lo:hi], rows[1:])
        if mode == "allreduce":
            if mine:
                _ordered_sum(rows[:1 + mine, :N], out=rows[0, :N])
            continue
        # ordered: rank j receives Gaussian slice j of every view of the round, in sequence order, and folds them into row 0
        send = rows[1:1 + mine].view(mine, world, chunk).permute(1, 0, 2).contiguous().view(world * mine, chunk)
        total = sum(sizes)
        timed(dist.all_to_all_single, recv[1:1 + total], send, output_split_sizes=sizes, input_split_sizes=[mine] * world, group=group)
        _ordered_sum(recv[:1 + total], out=recv[0])
    count_sum = runner.count_sum()
    timed(dist.all_reduce, count_sum, op=dist.ReduceOp.SUM, group=group)
    if mode == "allreduce":
        score = rows[0, :N].contiguous()
        timed(dist.all_reduce, score, op=dist.ReduceOp.SUM, group=group)
        timed.finish(world)
        return count_sum, score
    gathered = torch.empty(world * chunk, **f32)
    timed(dist.all_gather_into_tensor, gathered, recv[0].contiguous(), group=group)
    timed.finish(world)
    return count_sum, gathered[:N].contiguous()


# ------------------------------------------------------------------------------------------------------------------------
# compaction after the prune (scene/gaussian_model.py:564-600)

def compact_tensors(tensors, keep):
    """[t[keep] for t in tensors] -- every tensor [N, ...] with the same N, keep a bool/uint8 [N] mask -- as ONE prefix scan of
    the mask and ONE launch that moves the rows of all tensors (lg_compact_plan / lg_compact_rows), with a single 4-byte
    read-back (the number of kept rows, which sizes the outputs); torch's boolean indexing runs nonzero() + a host sync per
    tensor.  Row order is preserved: results equal t[keep] bit for bit.  HIP tensors only (no fallback); tensors whose rows
    are not a multiple of 4 bytes are indexed by torch with the same destination map."""
    import ctypes as C
    from . import _lib
    tensors = list(tensors)
    if not tensors:
        return []
    N = tensors[0].shape[0]
    dev = tensors[0].device
    if dev.type != "cuda":
        raise RuntimeError("compact_tensors runs on the MI355X HIP library only (no CPU fallback)")
    if any(t.shape[0] != N or t.device != dev for t in tensors) or keep.shape[0] != N:
        raise ValueError("compact_tensors: all tensors and the mask must share the first dimension and the device")
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    keep8 = keep.reshape(-1).to(device=dev, dtype=torch.uint8).contiguous()
    dest = torch.empty(N, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(lib.lg_compact_scratch_bytes(N), dtype=torch.uint8, device=dev)
    _lib.check(lib.lg_compact_plan(N, keep8.data_ptr(), dest.data_ptr(), count.data_ptr(), scratch.data_ptr(), stream))
    n_keep = int(count.item())                       # the one host read of the whole compaction
    srcs = [t.detach().contiguous() for t in tensors]
    outs = [torch.empty((n_keep,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in srcs]
    native = [i for i, t in enumerate(srcs) if N > 0 and t[0].numel() * t.element_size() % 4 == 0 and t[0].numel() > 0]
    for i in range(len(srcs)):
        if i not in native and N > 0 and srcs[i][0].numel() > 0:
            outs[i] = srcs[i][keep8.bool()]
    for lo in range(0, len(native), 32):
        part = native[lo:lo + 32]
        n = len(part)
        src_arr = (C.c_void_p * n)(*[srcs[i].data_ptr() for i in part])
        dst_arr = (C.c_void_p * n)(*[outs[i].data_ptr() if n_keep else srcs[i].data_ptr() for i in part])
        rb_arr = (C.c_int32 * n)(*[srcs[i][0].numel() * srcs[i].element_size() for i in part])
        if n_keep:
            _lib.check(lib.lg_compact_rows(N, dest.data_ptr(), n, src_arr, dst_arr, rb_arr, stream))
    return outs


def prune_points(model, mask):
    """GaussianModel.prune_points(mask) (scene/gaussian_model.py:584-600) including the optimizer surgery of _prune_optimizer
    (:564-582): the six parameters, both Adam moments of each and xyz_gradient_accum / denom / max_radii2D are compacted with
    valid = ~mask in ONE compact_tensors call instead of 21 boolean-index kernels.  `model` is a reference GaussianModel (or
    anything with the same attributes: optimizer with one named parameter per group, _xyz ... _rotation, the three
    bookkeeping tensors).  Same state afterwards as the reference leaves, bit for bit (tests/test_gpu_compact.py)."""
    from torch import nn
    valid = ~mask.reshape(-1).bool()
    opt = model.optimizer
    plan, tensors = [], []
    for group in opt.param_groups:
        p = group["params"][0]
        st = opt.state.get(p, None)
        plan.append((group, p, st, len(tensors)))
        tensors.append(p.data)
        if st is not None:
            tensors += [st["exp_avg"], st["exp_avg_sq"]]
    extra0 = len(tensors)
    tensors += [model.xyz_gradient_accum, model.denom, model.max_radii2D]
    outs = compact_tensors(tensors, valid)
    optimizable = {}
    for group, p, st, i in plan:
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = outs[i + 1], outs[i + 2]
            del opt.state[p]
            group["params"][0] = nn.Parameter(outs[i].requires_grad_(True))
            opt.state[group["params"][0]] = st
        else:
            group["params"][0] = nn.Parameter(outs[i].requires_grad_(True))
        optimizable[group["name"]] = group["params"][0]
    model._xyz = optimizable["xyz"]
    model._features_dc = optimizable["f_dc"]
    model._features_rest = optimizable["f_rest"]
    model._opacity = optimizable["opacity"]
    model._scaling = optimizable["scaling"]
    model._rotation = optimizable["rotation"]
    model.xyz_gradient_accum, model.denom, model.max_radii2D = outs[extra0], outs[extra0 + 1], outs[extra0 + 2]
    return optimizable


def prune_gaussians(model, percent, import_score):
    """GaussianModel.prune_gaussians (scene/gaussian_model.py:776-782): threshold by one radix select, then prune_points."""
    score = import_score.reshape(-1)
    if score.is_cuda:
        _, mask = _select_mask(score.detach().contiguous().float(), int(percent * (score.shape[0] - 1)))
        mask = mask.bool()
    else:
        mask = prune_mask(percent, import_score)
    prune_points(model, mask)
    return mask
