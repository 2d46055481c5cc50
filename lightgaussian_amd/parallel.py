"""Data-parallel helpers for training-style callers.

One process per GPU, Gaussians replicated, every rank renders a different camera, and the parameter
gradients are summed/averaged with RCCL before the optimizer step.  All six gradient tensors become ready at
the same moment (they are produced by one kernel, K9), so there is nothing to overlap inside a step; what
matters on xGMI -- point-to-point links, ring collectives bound by one link -- is FEW, LARGE collectives:
the gradients are flattened into buckets of `bucket_bytes` (default 512 MiB, i.e. one or two collectives at
6M Gaussians) instead of one all-reduce per tensor.
"""
import torch
import torch.distributed as dist


def shard_views(num_views, world_size, rank):
    """View indices rendered by `rank` when `num_views` cameras are dealt round-robin (training order)."""
    return list(range(rank, num_views, world_size))


def allreduce_gradients(params, group=None, average=True, bucket_bytes=512 << 20):
    """In-place all-reduce of `.grad` of every parameter that has one (same set on every rank), bucketed.
    Returns the number of collectives issued."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    buckets, cur, cur_bytes = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or g.dtype != cur[0].dtype):
            buckets.append(cur); cur, cur_bytes = [], 0
        cur.append(g); cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    works = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b]) if len(b) > 1 else b[0].reshape(-1)
        works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for w, flat, b in works:
        w.wait()
        if average:
            flat.div_(world)
        if len(b) > 1 or flat.data_ptr() != b[0].data_ptr():
            off = 0
            for g in b:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
    return len(works)


def make_student(teacher, sh_degree):
    """What distill_train.py:78-79 + GaussianModel.onedownSHdegree (scene/gaussian_model.py:129-136) produce: the same
    Gaussians with _features_rest cut to (sh_degree+1)^2 - 1 coefficients and active/max degree lowered."""
    from .synthetic import SyntheticGaussians
    keep = (sh_degree + 1) ** 2 - 1
    return SyntheticGaussians(teacher._xyz.detach().clone(), teacher._features_dc.detach().clone(),
                              teacher._features_rest[:, :keep].detach().clone(), teacher._scaling.detach().clone(),
                              teacher._rotation.detach().clone(), teacher._opacity.detach().clone(), sh_degree, sh_degree)
