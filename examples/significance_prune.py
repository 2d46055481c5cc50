#!/usr/bin/env python3
"""LightGaussian's prune step on MI355X, end to end, on a synthetic scene:

    count_render over all training views  ->  Global Significance score  ->  volume weighting  ->  prune mask

Single GPU:   python examples/significance_prune.py
8 GPUs:       python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/significance_prune.py

It is the code a user of the reference swaps in for `prune.prune_list(...)` + `calculate_v_imp_score` +
`GaussianModel.prune_gaussians` (prune_finetune.py:213-225); with >1 rank the cameras are sharded and the counts /
scores are reduced with RCCL so that every rank ends up with the same, bit-identical mask."""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver

import torch  # noqa: E402
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import synthetic as syn  # noqa: E402
from lightgaussian_amd.prune import prune_epilogue, prune_list_sharded  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-gaussians", type=int, default=1_000_000)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--prune-percent", type=float, default=0.66)   # scripts/run_prune_finetune.sh:37-47
    ap.add_argument("--v-pow", type=float, default=0.1)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    gaussians = syn.make_gaussians(args.n_gaussians).to(dev)
    cameras = [syn.orbit_camera(k, args.views, args.width, args.height).to(dev) for k in range(args.views)]
    background = torch.zeros(3, device=dev)
    with torch.no_grad():   # one-time costs (code object load, allocator growth, RCCL communicator) stay out of the timing
        prune_list_sharded(gaussians, cameras[:max(world, 2)], syn.PipelineParams(), background)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        gaussian_list, imp_list = prune_list_sharded(gaussians, cameras, syn.PipelineParams(), background)
        # calculate_v_imp_score + the prune_gaussians mask in one device-resident pass (two radix selects, no host read-back)
        v_list, mask, _ = prune_epilogue(gaussians, imp_list, args.v_pow, args.prune_percent)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if (not dist.is_initialized()) or dist.get_rank() == 0:
        print(f"{args.views} views x {args.n_gaussians} Gaussians on {world} GPU(s): {dt * 1e3:.1f} ms "
              f"({args.views / dt:.0f} views/s); hits {int(gaussian_list.sum())}, never hit {int((gaussian_list == 0).sum())}, "
              f"pruned {int(mask.sum())} ({100.0 * mask.float().mean():.1f} %)")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
