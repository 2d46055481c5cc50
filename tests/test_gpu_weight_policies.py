"""-m gpu: the per-hit significance weights (weight_policy "alpha" / "alpha_t") -- first-class since round 6.

SURVEY.md section 2.2 marks the weight the fork adds per (pixel, Gaussian) hit as UNVERIFIED (the submodule is not on disk,
/root/reference/.gitmodules:6-8; README.md:44 only says the fork "get[s] the Global Significant Score"; call site
gaussian_renderer/__init__.py:209-218, consumer prune.py:144-155).  The default here is the paper's sigma_j; should the fork add
alpha or alpha T instead, these policies are the path.  Their definition (oracle/lg_oracle.c lgo_forward):
every hit's fp32 weight rounded to the nearest multiple of 2^-40, the multiples added as 64-bit integers, the per-view score =
that integer rounded once to fp32.  Integer addition is associative, so -- unlike the float atomics of a CUDA implementation --
the result does not depend on the order the hits arrive in: the tests below demand BIT-IDENTICAL counts and scores against the
oracle's sequential loop, between two runs, between the colour-carrying and the significance-only kernel variants, across the
capacity-bounded and exact forwards, and through the sharded pass at every world size.
"""
import math
import os

import numpy as np
import pytest
import torch

import common
import gpu_common
from common import syn
from oracle import oracle
from test_gpu_parity import CASES, _np, _scene

pytestmark = pytest.mark.gpu
POLICIES = {"alpha": oracle.W_ALPHA, "alpha_t": oracle.W_ALPHA_T}


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("pol", sorted(POLICIES))
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"N{c['N']}_{c['W']}x{c['H']}_d{c['deg']}")
def test_per_hit_weight_scores_are_bit_identical_to_the_oracle(case, pol):
    from lightgaussian_amd import rasterizer
    kw = _scene(case)
    ref = oracle.forward(count=True, weight_policy=POLICIES[pol], **_np(kw))
    assert ref.score.max() > 0
    with rasterizer.options(weight_policy=pol):
        a = gpu_common.hip_forward_backward(kw, count=True)
        b = gpu_common.hip_forward_backward(kw, count=True)            # run to run: same bits (no float atomics)
    with rasterizer.options(weight_policy=pol, skip_color_in_count=True):
        c = gpu_common.hip_forward_backward(kw, count=True)            # the significance-only kernel variant (no colour, no image)
    with rasterizer.options(weight_policy=pol, sync_free=False):
        d = gpu_common.hip_forward_backward(kw, count=True)            # exact forward (read-back of the instance count)
    for out in (a, b, c, d):
        assert np.array_equal(out["count"], ref.count)
        bad = np.flatnonzero(_bits(out["score"]) != _bits(ref.score))
        assert bad.size == 0, (f"{bad.size} scores differ from the oracle, first: Gaussian {bad[0]} hip {out['score'][bad[0]]!r} "
                               f"oracle {ref.score[bad[0]]!r} ({ref.count[bad[0]]} hits)")
    # the image of a count forward does not depend on the weight policy
    assert np.array_equal(_bits(a["color"]), _bits(ref.color))
    # sanity of the definition: the fixed-point score is the real-number sum of the fp32 weights to within n 2^-41 + one fp32 rounding;
    # ALPHA_T's scores sum to (image opacity) = sum over pixels of (1 - T_final) up to the pixels' stop-threshold remainder
    assert np.all(a["score"][ref.count == 0] == 0) and np.all(a["score"][ref.count > 0] > 0)


def test_one_and_opacity_policies_unchanged_and_named_policies_resolve():
    from lightgaussian_amd import _lib, rasterizer
    kw = _scene(CASES[1])
    ref_one = oracle.forward(count=True, weight_policy=oracle.W_ONE, **_np(kw))
    ref_op = oracle.forward(count=True, weight_policy=oracle.W_OPACITY, **_np(kw))
    with rasterizer.options(weight_policy="one"):
        one = gpu_common.hip_forward_backward(kw, count=True)
    with rasterizer.options(weight_policy=_lib.WEIGHT_OPACITY):
        op = gpu_common.hip_forward_backward(kw, count=True)
    assert np.array_equal(one["score"], ref_one.count.astype(np.float32)) and np.array_equal(_bits(one["score"]), _bits(ref_one.score))
    assert np.array_equal(_bits(op["score"]), _bits(ref_op.score))
    with pytest.raises(ValueError):
        rasterizer.options(weight_policy="alphaT")
    with pytest.raises(ValueError):
        rasterizer.set_option("weight_policy", 7)
    assert rasterizer.resolve_options()["weight_policy"] == _lib.WEIGHT_OPACITY


@pytest.mark.parametrize("pol", sorted(POLICIES))
def test_long_lists_and_saturating_piles_keep_the_scores_bit_identical(pol):
    """Tile lists of several segments (segment_length 64) with saturating pixels: the float policies always take the serial walk; batches
    of 64 entries with every fill level of the 8-entry transposition chunks."""
    from lightgaussian_amd import rasterizer
    g = syn.make_gaussians(6000, seed=21, log_scale_mean=math.log(0.08), opacity_mean=1.5, extent=(1.0, 0.6, 1.0), log_scale_std=0.7)
    cam = syn.orbit_camera(2, 9, 208, 144, radius=4.0)
    kw = common.scene_kwargs(g, cam, 208, 144, deg=1, bg=(0.0, 0.0, 0.0), as_torch=True)
    ref = oracle.forward(count=True, weight_policy=POLICIES[pol], **_np(kw))
    assert ref.num_rendered / ((208 // 16) * (144 // 16)) > 300
    for extra in ({}, {"segment_length": 64}, {"segment_length": 64, "count_long_tiles": "parallel", "skip_color_in_count": True}):
        with rasterizer.options(weight_policy=pol, **extra):
            out = gpu_common.hip_forward_backward(kw, count=True)
        assert np.array_equal(out["count"], ref.count), extra
        assert np.array_equal(_bits(out["score"]), _bits(ref.score)), extra


@pytest.mark.parametrize("pol", sorted(POLICIES))
def test_sharded_pass_with_per_hit_weights_equals_the_reference_loop_over_oracle_views(pol):
    """prune_list_sharded(weight_policy=...) on the device (4 views in flight, sync-free forwards, significance-only kernels) against the
    reference's loop (prune.py:133-157: pop() from the end, in-place +=) over the ORACLE's per-view outputs: counts equal, scores and
    the prune mask bit-identical."""
    from lightgaussian_amd import prune as lg_prune
    dev = torch.device("cuda:0")
    N, W, H, V = 4000, 160, 96, 9
    g = syn.make_gaussians(N, seed=13, log_scale_mean=math.log(0.04), opacity_mean=0.5, extent=(2, 1.2, 2))
    cams = [syn.orbit_camera(k, V, W, H, radius=5.0) for k in range(V)]
    cnt_ref, imp_ref = None, None
    pc = g.to(dev)
    # the oracle is fed what the kernels see: the getters evaluated by torch ON THE DEVICE (torch's CPU exp / sigmoid differ from the
    # device's in the last bit; alpha -- unlike the hit counts of this scene -- notices)
    from test_gpu_full_size import _activated_on_device, _oracle_kw
    act = _activated_on_device(pc)
    for cam in cams[::-1]:
        f = oracle.forward(count=True, weight_policy=POLICIES[pol], **_oracle_kw(act, cam, W, H, 3, np.zeros(3)))
        if cnt_ref is None:
            cnt_ref, imp_ref = f.count.astype(np.int64), f.score.copy()
        else:
            cnt_ref += f.count
            imp_ref += f.score                      # float32 in-place adds in the reference's order
    with torch.no_grad():
        for streams, block in ((4, 24), (1, 2), (3, 4)):
            cnt, imp = lg_prune.prune_list_sharded(pc, [c.to(dev) for c in cams], syn.PipelineParams(), torch.zeros(3, device=dev),
                                                   weight_policy=pol, streams=streams, block=block)
            assert np.array_equal(cnt.cpu().numpy(), cnt_ref)
            assert np.array_equal(_bits(imp.cpu().numpy()), _bits(imp_ref)), (streams, block)
        m_ref = lg_prune.prune_mask(0.66, lg_prune.calculate_v_imp_score(g, torch.from_numpy(imp_ref), 0.1))
        m_hip = lg_prune.prune_mask(0.66, lg_prune.calculate_v_imp_score(g, imp.cpu(), 0.1))
    assert torch.equal(m_ref, m_hip)


def test_bench_count_mode_with_alpha_t_on_two_and_three_ranks_sharing_the_gpu():
    """`bench.py --mode count --weight-policy alpha_t --gpus W` with W processes on this box's one GPU (collectives over gloo): the sharded
    pass gives the counts, the ordered scores and the prune mask of the single-rank pass bit for bit -- the per-view fixed-point scores
    are pure functions of the view, so the partition of the cameras over ranks cannot show."""
    from test_gpu_round5 import _bench_ranks_on_one_gpu
    for world in (2, 3):
        c = _bench_ranks_on_one_gpu(world, "--mode", "count", "--weight-policy", "alpha_t", "--steps", "5", "--warmup", "1", "--views", str(5 * world))["significance_pass"]
        assert c["weight_policy"] == "alpha_t" and c["rccl_world_size"] == world and c["mask_identical_on_all_ranks"] is True
        assert c["mask_equals_1gpu"] is True and c["counts_equal_1gpu"] is True and c["scores_bit_identical_1gpu"] is True and c["hits"] > 0


def test_images_beyond_the_q24_40_range_are_refused_for_per_hit_weights():
    """2^24 pixels is what a per-view Q24.40 sum can hold (one Gaussian collects at most 0.99 per pixel): larger images are an error for the
    per-hit policies, not a silent overflow; the integer-derived default takes them."""
    from lightgaussian_amd import rasterizer
    g = syn.make_gaussians(50, seed=3, log_scale_mean=math.log(0.05))
    W, H = 4112, 4096                                     # 16.8 M pixels > 2^24
    cam = syn.orbit_camera(0, 4, W, H)
    kw = common.scene_kwargs(g, cam, W, H, deg=3, as_torch=True)
    with rasterizer.options(weight_policy="alpha_t"):
        with pytest.raises(Exception, match="2\\^24"):
            gpu_common.hip_forward_backward(kw, count=True)
    out = gpu_common.hip_forward_backward(kw, count=True)
    assert out["count"].sum() > 0


def test_score_out_and_count_sum_options_write_where_the_caller_says():
    """The fused epilogue of the significance pass (lg_view.count_sum, rasterizer options score_out / count_sum): the forward writes
    important_score INTO the caller's row and ADDS the view's hit count to the caller's running sum -- the `gaussian_list += ...;
    imp_list += ...` bookkeeping of prune.py:144-155 without a launch of its own.  Equal to the plain outputs for the integer and the
    per-hit policies; wrong shapes / dtypes are refused."""
    from lightgaussian_amd import rasterizer
    from lightgaussian_amd.gaussian_renderer import count_render
    dev = torch.device("cuda:0")
    N, W, H = 5000, 192, 128
    pc = syn.make_gaussians(N, seed=3, log_scale_mean=math.log(0.03)).to(dev)
    cams = [syn.orbit_camera(k, 5, W, H, radius=5.0).to(dev) for k in range(3)]
    bg, pipe = torch.zeros(3, device=dev), syn.PipelineParams()
    for pol in ("opacity", "alpha_t"):
        rows = torch.full((3, N + 7), -1.0, device=dev)               # (a wider matrix: the row slice is what the forward gets)
        running = torch.zeros(N, dtype=torch.int32, device=dev)
        plain_cnt = torch.zeros(N, dtype=torch.int32, device=dev)
        with torch.no_grad():
            for k, cam in enumerate(cams):
                ref = count_render(cam, pc, pipe, bg, options={"weight_policy": pol, "skip_color_in_count": True})
                plain_cnt += ref["gaussians_count"]
                out = count_render(cam, pc, pipe, bg, options={"weight_policy": pol, "skip_color_in_count": True,
                                                               "score_out": rows[k, :N], "count_sum": running})
                assert out["important_score"].data_ptr() == rows[k].data_ptr()
                assert torch.equal(out["gaussians_count"], ref["gaussians_count"])
                assert torch.equal(rows[k, :N].view(torch.int32), ref["important_score"].view(torch.int32))
        assert torch.equal(running, plain_cnt) and bool((rows[:, N:] == -1.0).all())
    with pytest.raises(ValueError):
        count_render(cams[0], pc, pipe, bg, options={"score_out": torch.zeros(N + 1, device=dev)})
    with pytest.raises(ValueError):
        count_render(cams[0], pc, pipe, bg, options={"count_sum": torch.zeros(N, dtype=torch.int64, device=dev)})
