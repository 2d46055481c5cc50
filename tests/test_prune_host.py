"""Host logic of the significance pass: prune_list order semantics, the sharded (multi-process) form over
gloo at world_size 2 and 3 against the single-process loop -- bit-identical scores and masks.  The
per-view renderer is replaced by the CPU oracle (count_fn hook): the GPU is not needed to test the
sharding / collective logic.  No GPU."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common
from common import syn
from lightgaussian_amd import prune as lg_prune
from oracle import oracle

N, W, H, V = 1500, 96, 64, 7


def _scene():
    g = syn.make_gaussians(N, seed=13, log_scale_mean=math.log(0.04), opacity_mean=0.5, extent=(2, 1.2, 2))
    cams = [syn.orbit_camera(k, V, W, H, radius=5.0) for k in range(V)]
    return g, cams


def oracle_count_fn(cam, pc, pipe, bg, weight_policy=oracle.W_OPACITY):
    """count_render() stand-in running the CPU oracle (test infrastructure)."""
    kw = common.scene_kwargs(pc, cam, cam.image_width, cam.image_height, bg=tuple(bg.tolist()))
    f = oracle.forward(count=True, weight_policy=weight_policy, **kw)
    return {"gaussians_count": torch.from_numpy(f.count.copy()), "important_score": torch.from_numpy(f.score.copy())}


def oracle_count_fn_alpha_t(cam, pc, pipe, bg):
    return oracle_count_fn(cam, pc, pipe, bg, oracle.W_ALPHA_T)


def oracle_count_fn_alpha(cam, pc, pipe, bg):
    return oracle_count_fn(cam, pc, pipe, bg, oracle.W_ALPHA)


COUNT_FNS = {"opacity": oracle_count_fn, "alpha": oracle_count_fn_alpha, "alpha_t": oracle_count_fn_alpha_t}


def test_prune_list_is_reference_loop_order():
    g, cams = _scene()
    bg = torch.zeros(3)
    cnt, imp = lg_prune.prune_list(g, cams, syn.PipelineParams(), bg, count_fn=oracle_count_fn)
    # reference loop restated literally (prune.py:133-157): pop() from the end, first view is the accumulator
    stack = list(cams)
    first = oracle_count_fn(stack.pop(), g, None, bg)
    c2, s2 = first["gaussians_count"], first["important_score"]
    while stack:
        p = oracle_count_fn(stack.pop(), g, None, bg)
        c2 += p["gaussians_count"]; s2 += p["important_score"]
    assert torch.equal(cnt, c2) and torch.equal(imp, s2)
    assert cnt.dtype == torch.int32 and imp.dtype == torch.float32
    # order matters in float: the reversed order gives (slightly) different bits somewhere
    rev = lg_prune.prune_list(g, cams[::-1], syn.PipelineParams(), bg, count_fn=oracle_count_fn)[1]
    assert torch.allclose(rev, imp, rtol=1e-5)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, out_dir, block=24, pol="opacity"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, cams = _scene()
        fn = COUNT_FNS[pol]
        cnt, imp = lg_prune.prune_list_sharded(g, cams, syn.PipelineParams(), torch.zeros(3), mode=mode, count_fn=fn, block=block, weight_policy=pol)
        if rank == 0:   # local_only inside a distributed job = the single-process loop (what bench.py's mask_equals_1gpu recomputes)
            c1, i1 = lg_prune.prune_list_sharded(g, cams, syn.PipelineParams(), torch.zeros(3), count_fn=fn, local_only=True, block=3, weight_policy=pol)
            assert torch.equal(c1, cnt) and (mode != "ordered" or torch.equal(i1, imp))
        v = lg_prune.calculate_v_imp_score(g, imp, 0.1)
        mask = lg_prune.prune_mask(0.66, v)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), cnt=cnt.numpy(), imp=imp.numpy(), mask=mask.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,block", [(2, "ordered", 24), (3, "ordered", 2), (2, "ordered", 1), (2, "allreduce", 3)])
def test_sharded_prune_pass_matches_single_process(world, mode, block, tmp_path):
    """block = views per rank per round: 24 -> one round; 2 at world 3 -> rounds of 6 views with a ragged last round in which
    two ranks own NO view (zero-length all_to_all splits); 1 -> seven rounds."""
    g, cams = _scene()
    cnt1, imp1 = lg_prune.prune_list(g, cams, syn.PipelineParams(), torch.zeros(3), count_fn=oracle_count_fn)
    mask1 = lg_prune.prune_mask(0.66, lg_prune.calculate_v_imp_score(g, imp1, 0.1))
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), block), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for o in outs:
        assert np.array_equal(o["cnt"], cnt1.numpy())                       # integer all-reduce: exact
        assert np.array_equal(o["cnt"], outs[0]["cnt"]) and np.array_equal(o["imp"].view(np.uint32), outs[0]["imp"].view(np.uint32))
        if mode == "ordered":
            assert np.array_equal(o["imp"].view(np.uint32), imp1.numpy().view(np.uint32)), "ordered mode must be bit-identical"
            assert np.array_equal(o["mask"], mask1.numpy())                 # prune-mask Hamming distance 0
        else:
            assert np.allclose(o["imp"], imp1.numpy(), rtol=1e-5)
            assert np.count_nonzero(o["mask"] != mask1.numpy()) <= 2


@pytest.mark.parametrize("pol", ["alpha", "alpha_t"])
@pytest.mark.parametrize("world,block", [(1, 24), (2, 24), (3, 2)])
def test_sharded_pass_with_per_hit_weights_is_world_size_independent(world, block, pol, tmp_path):
    """The per-hit weight policies (Q24.40 fixed-point per-view sums: pure functions of the view) through the same ordered exchange:
    scores bit-identical to the single-process reference loop and prune-mask Hamming distance 0 at world size 1, 2 and 3."""
    g, cams = _scene()
    fn = COUNT_FNS[pol]
    cnt1, imp1 = lg_prune.prune_list(g, cams, syn.PipelineParams(), torch.zeros(3), count_fn=fn)
    imp_op = lg_prune.prune_list(g, cams, syn.PipelineParams(), torch.zeros(3), count_fn=oracle_count_fn)[1]
    assert not torch.equal(imp1, imp_op)                                    # really another weight
    mask1 = lg_prune.prune_mask(0.66, lg_prune.calculate_v_imp_score(g, imp1, 0.1))
    if world == 1:
        cnt, imp = lg_prune.prune_list_sharded(g, cams, syn.PipelineParams(), torch.zeros(3), count_fn=fn, block=block, weight_policy=pol)
        assert torch.equal(cnt, cnt1) and torch.equal(imp, imp1)
        return
    mp.spawn(_worker, args=(world, _free_port(), "ordered", str(tmp_path), block, pol), nprocs=world, join=True)
    for r in range(world):
        o = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(o["cnt"], cnt1.numpy())
        assert np.array_equal(o["imp"].view(np.uint32), imp1.numpy().view(np.uint32))
        assert np.array_equal(o["mask"], mask1.numpy())


def test_weight_policy_names():
    from lightgaussian_amd import _lib, rasterizer
    assert [rasterizer.weight_policy_id(n) for n in ("one", "opacity", "alpha", "ALPHA_T")] == [_lib.WEIGHT_ONE, _lib.WEIGHT_OPACITY, _lib.WEIGHT_ALPHA, _lib.WEIGHT_ALPHA_T]
    assert rasterizer.weight_policy_id(_lib.WEIGHT_ALPHA) == _lib.WEIGHT_ALPHA
    for bad in ("alphaT", 4, -1):
        with pytest.raises(ValueError):
            rasterizer.weight_policy_id(bad)


def test_single_process_pass_in_bounded_chunks_equals_the_reference_loop():
    """The running-sum form (scratch O(block * N)) adds the views in the reference's order: same bits for every
    block size, including block > V and block = 1."""
    g, cams = _scene()
    cnt1, imp1 = lg_prune.prune_list(g, cams, syn.PipelineParams(), torch.zeros(3), count_fn=oracle_count_fn)
    for block in (1, 3, 7, 50):
        cnt, imp = lg_prune.prune_list_sharded(g, cams, syn.PipelineParams(), torch.zeros(3), count_fn=oracle_count_fn, block=block)
        assert torch.equal(cnt, cnt1) and torch.equal(imp, imp1), block


def test_round_schedule_covers_every_view_once_in_sequence_order():
    for Vv in (1, 7, 200, 13):
        for w in (1, 2, 3, 8):
            for block in (1, 4, 24):
                seen = []
                for t in range(lg_prune.num_rounds(Vv, w, block)):
                    for r in range(w):
                        lo, hi = lg_prune.round_views(Vv, w, r, block, t)
                        assert hi - lo <= block
                        seen += list(range(lo, hi))
                assert seen == list(range(Vv))


def test_shard_bounds_partition():
    for Vv in (1, 7, 200, 13):
        for w in (1, 2, 3, 8):
            spans = [lg_prune.shard_bounds(Vv, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == Vv
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_mask_threshold_semantics_ties_pruned():
    score = torch.tensor([0.0, 0.0, 0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0])
    m = lg_prune.prune_mask(0.1, score)      # index int(0.1*9)=0 -> thr 0.0 -> all three zeros pruned
    assert m.tolist() == [True, True, True] + [False] * 7
    m = lg_prune.prune_mask(0.66, score)     # index int(0.66*9)=5 -> thr 3.0
    assert m.sum().item() == 6


def test_select_based_epilogue_is_bit_identical_to_sort_based():
    g = torch.Generator().manual_seed(3)
    for n in (10, 1000, 50001):
        scal = torch.exp(torch.randn(n, 3, generator=g) * 0.8 - 4.0)
        imp = torch.rand(n, generator=g) * 100
        imp[torch.rand(n, generator=g) < 0.3] = 0.0

        class GM:
            get_scaling = scal
        for v_pow in (0.1, 0.5):
            a = lg_prune.calculate_v_imp_score(GM, imp, v_pow)
            b = lg_prune.calculate_v_imp_score_select(GM, imp, v_pow)
            assert torch.equal(a, b)
            for pct in (0.0, 0.1, 0.66, 1.0):
                assert torch.equal(lg_prune.prune_mask(pct, a), lg_prune.prune_mask_select(pct, a))
