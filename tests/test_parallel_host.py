"""Bucketed gradient all-reduce over gloo (world size 2).  No GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common  # noqa: F401
from lightgaussian_amd import parallel, synthetic as syn


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir, bucket_bytes):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        params = [torch.zeros(1000, 3, requires_grad=True), torch.zeros(1000, 8, 3, requires_grad=True), torch.zeros(1000, 1, requires_grad=True),
                  torch.zeros(7, requires_grad=True)]
        for p in params[:3]:
            p.grad = torch.randn(p.shape, generator=g)
        n = parallel.allreduce_gradients(params, bucket_bytes=bucket_bytes)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), n=n, **{f"g{i}": p.grad.numpy() for i, p in enumerate(params[:3])})
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_matches_manual_average(tmp_path):
    for bucket in (512 << 20, 40_000):     # one bucket / several buckets
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), bucket), nprocs=2, join=True)
        outs = [np.load(tmp_path / f"r{r}.npz") for r in range(2)]
        gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
        per_rank = [[torch.randn(s, generator=gens[r]) for s in [(1000, 3), (1000, 8, 3), (1000, 1)]] for r in range(2)]
        for i in range(3):
            mean = ((per_rank[0][i] + per_rank[1][i]) / 2).numpy()
            for o in outs:
                assert np.allclose(o[f"g{i}"], mean, rtol=1e-6, atol=1e-7)
        assert int(outs[0]["n"]) == (1 if bucket > 1 << 20 else 3)


def test_shard_views_and_student():
    assert parallel.shard_views(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((parallel.shard_views(13, 3, r) for r in range(3)), [])) == list(range(13))
    t = syn.make_gaussians(50, sh_degree=3)
    s = parallel.make_student(t, 2)
    assert s._features_rest.shape == (50, 8, 3) and s.active_sh_degree == 2 and s.get_features.shape == (50, 9, 3)
    assert torch.equal(s._features_rest, t._features_rest[:, :8])
