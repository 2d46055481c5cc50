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


def _overlap_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lightgaussian_amd import rasterizer
        g = torch.Generator().manual_seed(200 + rank)
        n = 1000
        grads = {"_xyz": torch.randn(n, 3, generator=g), "_features_rest": torch.randn(n, 15, 3, generator=g), "_opacity": torch.randn(n, 1, generator=g)}
        model = type("M", (), {})()
        for k in grads:
            setattr(model, k, torch.zeros_like(grads[k]).requires_grad_(True))
        with parallel.OverlappedGradAllReduce(chunks=4) as ar:
            hook = rasterizer._GRAD_CHUNKS["hook"]
            assert hook is not None and rasterizer._GRAD_CHUNKS["chunks"] == 4
            for first in range(0, n, 256):                       # what lg_backward_chunked reports: ranges of whole workgroups
                hook(first, min(256, n - first), grads)
        assert rasterizer._GRAD_CHUNKS["hook"] is None
        out = ar.finish(model)
        assert out["_xyz"] is grads["_xyz"] and model._xyz.grad is grads["_xyz"]
        np.savez(os.path.join(out_dir, f"o{rank}.npz"), **{k: v.numpy() for k, v in grads.items()})
    finally:
        dist.destroy_process_group()


def test_overlapped_allreduce_of_gradient_ranges(tmp_path):
    """The host logic of the overlapped data-parallel step: every Gaussian range reported by the chunked backward is
    all-reduced (one packed collective per range) and scattered back; finish() installs the averaged tensors as .grad."""
    mp.spawn(_overlap_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    outs = [np.load(tmp_path / f"o{r}.npz") for r in range(2)]
    gens = [torch.Generator().manual_seed(200 + r) for r in range(2)]
    per_rank = [{k: torch.randn(s, generator=gens[r]) for k, s in (("_xyz", (1000, 3)), ("_features_rest", (1000, 15, 3)), ("_opacity", (1000, 1)))}
                for r in range(2)]
    for k in ("_xyz", "_features_rest", "_opacity"):
        mean = ((per_rank[0][k] + per_rank[1][k]) / 2).numpy()
        for o in outs:
            assert np.allclose(o[k], mean, rtol=1e-6, atol=1e-7), k


def _visible_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(300 + rank)
        N = 1003
        vis = torch.rand(N, generator=g) < 0.4                      # what this rank's camera saw
        shapes = [(N, 3), (N, 1, 3), (N, 8, 3), (N, 1), (N, 4)]
        params = [torch.zeros(s, requires_grad=True) for s in shapes]
        for p in params:
            p.grad = torch.randn(p.shape, generator=g) * vis.view(-1, *([1] * (p.dim() - 1)))   # zero rows where nothing was seen
        dense = [p.grad.clone() for p in params]
        for d in dense:
            dist.all_reduce(d); d.div_(world)
        k, n = parallel.allreduce_gradients_visible(params, vis)
        union = vis.clone().to(torch.int32)
        dist.all_reduce(union)
        assert n == N and k == int((union > 0).sum())
        for p, d in zip(params, dense):
            assert torch.equal(p.grad, d)                            # world 2: a + b == b + a, bit for bit
        # nothing visible anywhere: no rows, gradients untouched
        for p in params:
            p.grad = torch.zeros_like(p)
        assert parallel.allreduce_gradients_visible(params, torch.zeros(N, dtype=torch.bool)) == (0, N)
        np.savez(os.path.join(out_dir, f"v{rank}.npz"), k=k)
    finally:
        dist.destroy_process_group()


def test_visible_rows_allreduce_equals_the_dense_one(tmp_path):
    """allreduce_gradients_visible: MAX of the visibility flags (the union), then one sum all-reduce of the union's rows only -- same
    gradients as the dense all-reduce (rows nobody saw are exactly zero everywhere)."""
    mp.spawn(_visible_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ks = [int(np.load(tmp_path / f"v{r}.npz")["k"]) for r in range(2)]
    assert ks[0] == ks[1] and 500 < ks[0] < 800                     # 1 - 0.6^2 = 64 % of 1003 rows
