"""-m gpu: round-5 kernels against what they replace.

  (a) lg_backward's rgb_only mode + lg_sh_grad_from_rgb (the rank-one SH-gradient exchange of the data-parallel step,
      parallel.RankOneSHExchange): the coefficient gradients rebuilt from dL/d(rgb) and the camera centre are the SAME BITS K9 writes
      for that view, every other gradient of the view is untouched, two views summed in view order equal autograd's accumulation,
      and the exchange object run through RCCL at world size 1 (collectives forced) returns the local result.
  (b) (round 5's prototype of K7 on the other parallel axis, lg_blend_bwd_splat, was removed in round 6: EXPERIMENTS.md, "K7 structure")
  (c) the pair step of the forward blend on scalar lane masks (fwd_pair_m): covered by every bit-parity test of the suite (counts,
      scores, canonical images are compared with the oracle bit for bit); here only the long-tile rewalk, which shares it."""
import math
import os

import numpy as np
import pytest
import torch

import common
import gpu_common
from common import syn
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RAW = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _scene(N=3000, seed=5, scale=0.03, active=3, stored=3, opm=0.0):
    g = syn.make_gaussians(N, sh_degree=stored, seed=seed, log_scale_mean=math.log(scale), opacity_mean=opm, extent=(2, 1.2, 2), rest_std=0.2)
    g.active_sh_degree = active
    return g


class _Sink:
    def __init__(self):
        self.views = []

    def add(self, drgb, campos, deg):
        self.views.append((drgb.clone(), campos.clone(), deg))


# ---- (a) -----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "literal"])
@pytest.mark.parametrize("D", [0, 1, 2, 3])
def test_sh_gradients_rebuilt_from_drgb_are_the_bits_k9_writes(D, fused):
    from lightgaussian_amd.gaussian_renderer import render
    from lightgaussian_amd import parallel
    dev = torch.device(DEV)
    W, H = 176, 112
    g = _scene(active=D)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    gimg = torch.from_numpy(np.random.RandomState(3).randn(3, H, W).astype(np.float32)).to(dev)
    cam = syn.orbit_camera(1, 7, W, H, radius=5.0).to(dev)
    res = {}
    for mode in ("dense", "rgb"):
        pc = g.to(dev).requires_grad_(True)
        sink = _Sink() if mode == "rgb" else None
        opts = {"fuse_getters": fused}
        if sink is not None:
            opts["sh_grad_sink"] = sink
        pkg = render(cam, pc, syn.PipelineParams(), bg, options=opts)
        (pkg["render"] * gimg).sum().backward()
        grads = {n: (getattr(pc, n).grad.clone() if getattr(pc, n).grad is not None else None) for n in RAW}
        if sink is not None:
            assert len(sink.views) == 1 and grads["_features_dc"] is None and grads["_features_rest"] is None
            drgb, cp, deg = sink.views[0]
            assert deg == D and torch.equal(cp.reshape(3), cam.camera_center.reshape(3))
            g_dc, g_rest = parallel.sh_grad_from_rgb(pc._xyz, cp.reshape(1, 3), drgb.unsqueeze(0), D, 16)
            grads["_features_dc"], grads["_features_rest"] = g_dc, g_rest
        res[mode] = (pkg["render"].detach().clone(), grads)
    assert torch.equal(res["dense"][0], res["rgb"][0])
    for n in RAW:
        a, b = res["dense"][1][n], res["rgb"][1][n]
        assert torch.equal(a, b.view_as(a)), f"{n}: rebuilt from dRGB differs from K9's own rows (D={D}, fused={fused})"
    assert float(res["dense"][1]["_features_dc"].abs().sum()) > 0
    if D > 0:
        assert float(res["dense"][1]["_features_rest"][:, :(D + 1) ** 2 - 1].abs().sum()) > 0
    assert float(res["dense"][1]["_features_rest"][:, (D + 1) ** 2 - 1:].abs().sum()) == 0.0


@pytest.mark.parametrize("stored", [0, 1, 2])
def test_sh_gradients_rebuilt_from_drgb_at_lower_stored_degrees(stored):
    """M = 1, 4, 9 stored coefficients (a student after onedownSHdegree; a degree-0 model has no _features_rest rows at all)."""
    from lightgaussian_amd.gaussian_renderer import render
    from lightgaussian_amd import parallel
    dev = torch.device(DEV)
    W, H = 128, 80
    g = _scene(N=1500, active=stored, stored=stored, seed=31)
    M = (stored + 1) ** 2
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gimg = torch.from_numpy(np.random.RandomState(8).randn(3, H, W).astype(np.float32)).to(dev)
    cam = syn.orbit_camera(4, 7, W, H, radius=5.0).to(dev)
    pc = g.to(dev).requires_grad_(True)
    (render(cam, pc, syn.PipelineParams(), bg)["render"] * gimg).sum().backward()
    sink = _Sink()
    pc2 = g.to(dev).requires_grad_(True)
    (render(cam, pc2, syn.PipelineParams(), bg, options={"sh_grad_sink": sink})["render"] * gimg).sum().backward()
    drgb, cp, deg = sink.views[0]
    g_dc, g_rest = parallel.sh_grad_from_rgb(pc2._xyz, cp.reshape(1, 3), drgb.unsqueeze(0), deg, M)
    assert deg == stored and g_rest.shape == (1500, M - 1, 3)
    assert torch.equal(g_dc, pc._features_dc.grad)
    if M > 1:
        assert torch.equal(g_rest, pc._features_rest.grad)
    for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(pc2, n).grad, getattr(pc, n).grad), n


def test_two_views_summed_in_view_order_equal_the_accumulated_dense_gradient():
    from lightgaussian_amd.gaussian_renderer import render
    from lightgaussian_amd import parallel
    dev = torch.device(DEV)
    W, H = 160, 96
    g = _scene(N=4101, seed=9)                       # N not a multiple of 64
    bg = torch.zeros(3, device=dev)
    cams = [syn.orbit_camera(k, 7, W, H, radius=5.0).to(dev) for k in (0, 3)]
    gimgs = [torch.from_numpy(np.random.RandomState(k).randn(3, H, W).astype(np.float32)).to(dev) for k in (1, 2)]
    pc = g.to(dev).requires_grad_(True)
    for cam, gi in zip(cams, gimgs):
        (render(cam, pc, syn.PipelineParams(), bg)["render"] * gi).sum().backward()
    dense = {n: getattr(pc, n).grad.clone() for n in RAW}
    ex = parallel.RankOneSHExchange()                # no process group: the views stay local
    pc2 = g.to(dev).requires_grad_(True)
    for cam, gi in zip(cams, gimgs):
        (render(cam, pc2, syn.PipelineParams(), bg, options={"sh_grad_sink": ex})["render"] * gi).sum().backward()
    g_dc, g_rest = ex.finish(pc2._xyz, 16)
    assert torch.equal(g_dc, dense["_features_dc"]) and torch.equal(g_rest, dense["_features_rest"])
    for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(pc2, n).grad, dense[n]), n
    # one call over both views == two calls with accumulate (what finish() did): V = 2 in one launch
    drgbs, cps = [], []
    s = _Sink()
    pc3 = g.to(dev).requires_grad_(True)
    for cam, gi in zip(cams, gimgs):
        (render(cam, pc3, syn.PipelineParams(), bg, options={"sh_grad_sink": s})["render"] * gi).sum().backward()
    drgb = torch.stack([v[0] for v in s.views]); cp = torch.stack([v[1].reshape(3) for v in s.views])
    one = parallel.sh_grad_from_rgb(pc3._xyz, cp, drgb, 3, 16, divisor=2.0)
    assert torch.equal(one[0], dense["_features_dc"] / 2.0) and torch.equal(one[1], dense["_features_rest"] / 2.0)


def test_rank_one_exchange_through_rccl_at_world_size_one():
    """The collective path of RankOneSHExchange (all_gather_into_tensor on a side stream behind K9) on the RCCL backend, forced at
    world size 1: same gradients as without a process group."""
    import subprocess
    import sys
    code = r'''
import os, math, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["LG_ROOT"], "tests")); sys.path.insert(0, os.environ["LG_ROOT"])
from common import syn
from lightgaussian_amd.gaussian_renderer import render
from lightgaussian_amd import parallel
torch.cuda.set_device(0); dev = torch.device("cuda:0")
dist.init_process_group("nccl", device_id=dev)
g = syn.make_gaussians(3000, sh_degree=3, seed=5, log_scale_mean=math.log(0.03), extent=(2, 1.2, 2), rest_std=0.2)
W, H = 160, 96
cams = [syn.orbit_camera(k, 7, W, H, radius=5.0).to(dev) for k in (0, 2)]
bg = torch.zeros(3, device=dev)
gi = torch.from_numpy(np.random.RandomState(1).randn(3, H, W).astype(np.float32)).to(dev)
out = []
for force in (False, True):
    pc = g.to(dev).requires_grad_(True)
    ex = parallel.RankOneSHExchange(force=force)
    for cam in cams:
        (render(cam, pc, syn.PipelineParams(), bg, options={"sh_grad_sink": ex})["render"] * gi).sum().backward()
    nbytes = ex.bytes_on_wire
    out.append(ex.finish(pc._xyz, 16) + (nbytes,))
assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
assert out[0][2] == 0 and out[1][2] == 2 * (3 * 3000 + 3) * 4
assert float(out[0][1].abs().sum()) > 0
dist.destroy_process_group()
print("RANK1_OK")
'''
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, LG_ROOT=common.ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RANK1_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


# ---- (b) -----------------------------------------------------------------------------------------------------------------------
def _grads(g, cam, W, H, bg, gimg, options, fused=True):
    from lightgaussian_amd.gaussian_renderer import render
    dev = torch.device(DEV)
    pc = g.to(dev).requires_grad_(True)
    pkg = render(cam.to(dev), pc, syn.PipelineParams(), bg, options=dict(options, fuse_getters=fused))
    (pkg["render"] * gimg).sum().backward()
    out = {n: getattr(pc, n).grad.detach().cpu().numpy() for n in RAW}
    out["means2D"] = pkg["viewspace_points"].grad.detach().cpu().numpy()
    return out


# ---- (d) parallel long-tile walk of the significance-only pass (lg_count_seg / _rewalk / _fixup) ---------------------------------------
def _count_pass(g, cam, W, H, options):
    """count_render as prune_list_sharded issues it (getters frozen, colours skipped) -> counts, scores, and the per-view meta words."""
    import ctypes as C
    from lightgaussian_amd import _lib, rasterizer
    from lightgaussian_amd.rasterizer import GaussianRasterizationSettings
    dev = torch.device(DEV)
    with torch.no_grad():
        t = dict(means3D=g.get_xyz.to(dev).contiguous(), opacities=g.get_opacity.to(dev).contiguous(), shs=g.get_features.to(dev).contiguous(),
                 scales=g.get_scaling.to(dev).contiguous(), rotations=g.get_rotation.to(dev).contiguous())
    camd = cam.to(dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                       camd.full_proj_transform, 3, camd.camera_center, False, False, True)
    opts = rasterizer.resolve_options(dict(options, skip_color_in_count=True, sync_free=False))
    call = rasterizer._Call(rs, t["means3D"], t["shs"], None, t["opacities"], t["scales"], t["rotations"], None, exact=True, opts=opts)
    lib = _lib.load()
    _color, radii, cnt, score, _geom, binning, _img, R = rasterizer._native_forward(lib, call, rs, True)
    meta = torch.zeros(16, dtype=torch.int32, device=dev)
    _lib.check(lib.lg_debug_view_meta(C.byref(call.view), binning.data_ptr(), int(R), meta.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return cnt.cpu().numpy(), score.cpu().numpy(), meta.cpu().numpy().view(np.uint32), t


COUNT_SCENES = [dict(N=5000, W=96, H=64, scale=0.12, opm=1.5, seed=17),        # opaque pile: pixels saturate inside the first segments
                dict(N=9000, W=160, H=96, scale=0.06, opm=-0.5, seed=6),       # semi-opaque, deep: pixels saturate inside later segments, many never
                dict(N=6000, W=130, H=70, scale=0.1, opm=0.5, seed=7)]


@pytest.mark.parametrize("wide", [False, True], ids=["band", "wide_band"])
@pytest.mark.parametrize("S", [64, 128])
@pytest.mark.parametrize("sc", COUNT_SCENES, ids=lambda s: f"N{s['N']}_op{s['opm']}")
def test_parallel_count_walk_gives_the_serial_counts_bit_for_bit(sc, S, wide):
    """Significance-only pass, every multi-segment list through lg_count_seg / _rewalk / _fixup (count_long_tiles="parallel"): hit counts and
    scores bit-identical to the serial walk and to the oracle.  wide_band: the comparison band 4096 x wider, so that the exact fix-up
    resolves hundreds of pixels instead of (usually) none -- the counts must not move."""
    g = _scene(N=sc["N"], seed=sc["seed"], scale=sc["scale"], opm=sc["opm"])
    W, H = sc["W"], sc["H"]
    cam = syn.orbit_camera(2, 7, W, H, radius=5.0)
    base = {"segment_length": S}
    c_ser, s_ser, m_ser, t = _count_pass(g, cam, W, H, dict(base, count_long_tiles="serial"))
    c_par, s_par, m_par, _ = _count_pass(g, cam, W, H, dict(base, count_long_tiles="parallel", count_wide_band=wide))
    # (the serial pass launches no work-list workgroup: its meta words are whatever the allocator left there)
    assert m_par[4] > 0 and m_par[2] == S and m_par[1] > 2 * S, m_par[:6]   # the parallel kernels really had items, of lists of several segments
    assert np.array_equal(c_ser, c_par), f"{int((c_ser != c_par).sum())} counts differ, sum {int(c_ser.sum())} vs {int(c_par.sum())}, fix-ups {m_par[5]}"
    assert np.array_equal(s_ser.view(np.uint32), s_par.view(np.uint32))
    if wide:
        assert m_par[5] > 0, "the wide band sent no pixel through the exact fix-up: the test does not exercise it"
    kw = dict({k: v.cpu().numpy() for k, v in t.items()}, W=W, H=H, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=np.zeros(3, np.float32),
              viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), sh_degree=3)
    ref = oracle.forward(count=True, **kw)
    assert np.array_equal(c_par, ref.count) and np.array_equal(s_par.view(np.uint32), ref.score.view(np.uint32))
    assert int(ref.count.sum()) > 10000


def test_parallel_count_walk_on_the_heavy_tailed_scene():
    """count_long_tiles="parallel" with the default segment length on a heavy-tailed scene (lists of thousands of entries): counts equal the
    serial walk's; the default rule keeps the significance pass serial (measured faster with views in flight), and
    count_render -- which returns an image -- always walks serially."""
    from lightgaussian_amd.gaussian_renderer import count_render
    g = syn.make_gaussians(400_000, seed=9)
    syn.make_heavy_tailed(g, frac=0.08)
    W, H = 960, 540
    cam = syn.orbit_camera(0, 10, W, H)
    c_ser, s_ser, _m, _ = _count_pass(g, cam, W, H, dict(count_long_tiles="serial"))
    c_par, s_par, m_par, _ = _count_pass(g, cam, W, H, dict(count_long_tiles="parallel"))
    c_def, s_def, _m2, _ = _count_pass(g, cam, W, H, {})
    assert m_par[4] > 0 and m_par[1] > 4 * 512, m_par[:6]
    assert np.array_equal(c_ser, c_par) and np.array_equal(s_ser.view(np.uint32), s_par.view(np.uint32))
    assert np.array_equal(c_ser, c_def)
    dev = torch.device(DEV)
    with torch.no_grad():
        pkg = count_render(cam.to(dev), g.to(dev), syn.PipelineParams(), torch.zeros(3, device=dev))
    assert np.array_equal(pkg["gaussians_count"].cpu().numpy(), c_ser)



# ---- (e) the data-parallel step of bench.py at world size 2 on ONE GPU (collectives over gloo on device tensors) -----------------------
def _bench_two_ranks(tmp_path, *extra):
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LG_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--n-gaussians", "150000", "--width", "480",
           "--height", "272", "--views", "8", "--scale", "0.02", "--no-cpu-baseline", "--no-roofline", "--no-literal", "--no-c4-leg", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=common.ROOT, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_data_parallel_step_with_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` for real -- two processes, the real kernels, the per-rank camera shard, the rank-one SH exchange behind K9, the
    dense all-reduce of the other four tensors -- on the one GPU of this box (LG_BENCH_SHARE_GPU=1: both ranks on cuda:0, collectives over
    gloo).  Every rank must end a step with the same gradients; the rank-one exchange must give the gradients of the dense exchange BIT
    FOR BIT (two ranks: (t0 + t1) / 2 either way); a camera batch per rank and the overlapped all-reduce must run."""
    r1 = _bench_two_ranks(tmp_path)
    dn = _bench_two_ranks(tmp_path, "--dense-allreduce")
    assert r1["n_gpus"] == 2 and "test_mode" in r1 and r1["data_parallel"]["views_per_rank_per_step"] == 1
    assert r1["gradients_identical_on_all_ranks"] is True and dn["gradients_identical_on_all_ranks"] is True
    assert r1["gradient_sha256"] == dn["gradient_sha256"], "rank-one SH exchange and dense all-reduce left different gradients"
    assert 0 < r1["data_parallel"]["bytes_on_wire_per_step"] < 0.5 * dn["data_parallel"]["bytes_on_wire_per_step"]
    ov = _bench_two_ranks(tmp_path, "--dp-overlap")
    assert ov["gradients_identical_on_all_ranks"] is True and ov["gradient_sha256"] == r1["gradient_sha256"], (ov["gradient_sha256_per_tensor"], r1["gradient_sha256_per_tensor"])
    kv = _bench_two_ranks(tmp_path, "--views-per-rank", "2")
    assert kv["gradients_identical_on_all_ranks"] is True and kv["data_parallel"]["views_per_rank_per_step"] == 2
    vis = _bench_two_ranks(tmp_path, "--visible-allreduce")
    assert vis["gradients_identical_on_all_ranks"] is True and vis["gradient_sha256"] == dn["gradient_sha256"]


def _bench_ranks_on_one_gpu(world, *extra):
    """`bench.py --gpus world` as `world` processes on this box's one GPU (LG_BENCH_SHARE_GPU=1, collectives over gloo) -> its JSON line."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LG_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(common.ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--n-gaussians", "150000", "--width", "480", "--height", "272",
           "--scale", "0.02", "--no-cpu-baseline", "--no-roofline", "--no-literal", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=common.ROOT, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_c4_pass_with_two_ranks_on_one_gpu(tmp_path):
    """BASELINE configs[3] in miniature, for real at world size 2 (two processes on this box's one GPU, collectives over gloo): the sharded
    significance pass of `bench.py --mode count` (int32 count all-reduce, round-wise ordered all_to_all of the scores, all_gather) and the
    C4 leg of the default line: counts, ordered scores and the prune mask equal to the single-rank pass recomputed on rank 0."""
    def run(*extra):
        return _bench_ranks_on_one_gpu(2, *extra)
    c = run("--mode", "count", "--steps", "7", "--warmup", "2", "--views", "14")["significance_pass"]
    assert c["views"] == 14 and c["rccl_world_size"] == 2 and c["mask_identical_on_all_ranks"] is True
    assert c["mask_equals_1gpu"] is True and c["counts_equal_1gpu"] is True and c["scores_bit_identical_1gpu"] is True and c["hits"] > 0
    d = run("--steps", "3", "--warmup", "1", "--views", "8")          # the default line at N > 1: data-parallel step + the C4 leg
    c4 = d["c4_significance_pass"]
    assert c4["views"] == 200 and c4["views_per_rank"] == 100 and c4["mask_identical_on_all_ranks"] is True
    assert c4["mask_equals_1gpu"] is True and c4["counts_equal_1gpu"] is True and c4["scores_bit_identical_1gpu"] is True
    assert d["gradients_identical_on_all_ranks"] is True


def test_bench_with_four_ranks_on_one_gpu(tmp_path):
    """The same at world size FOUR (the mask must equal the 1-GPU mask at 1 / 2 / 4 / 8 ranks): four processes, the real kernels,
    the round-wise all_to_all of the score rows with three peers, 7 views per rank; and the data-parallel step with three peers per
    all-gather -- every rank ends the step with the same gradient bits, with the rank-one SH exchange and with the dense all-reduce (at
    four ranks the two associate the sum differently: equal to rounding, not bit for bit)."""
    c = _bench_ranks_on_one_gpu(4, "--mode", "count", "--steps", "7", "--warmup", "1", "--views", "28")["significance_pass"]
    assert c["views"] == 28 and c["rccl_world_size"] == 4 and c["mask_identical_on_all_ranks"] is True
    assert c["mask_equals_1gpu"] is True and c["counts_equal_1gpu"] is True and c["scores_bit_identical_1gpu"] is True and c["hits"] > 0
    d = _bench_ranks_on_one_gpu(4, "--steps", "3", "--warmup", "1", "--views", "8", "--no-c4-leg")
    assert d["n_gpus"] == 4 and d["gradients_identical_on_all_ranks"] is True and d["data_parallel"]["exchange"].startswith("SH gradients rebuilt")
    assert d["data_parallel"]["rccl_world_size"] == 4
    dn = _bench_ranks_on_one_gpu(4, "--steps", "3", "--warmup", "1", "--views", "8", "--no-c4-leg", "--dense-allreduce")
    assert dn["gradients_identical_on_all_ranks"] is True and dn["data_parallel"]["exchange"].startswith("dense tensors")
    assert 0 < d["data_parallel"]["bytes_on_wire_per_step"] < 0.5 * dn["data_parallel"]["bytes_on_wire_per_step"]


def test_bench_with_eight_ranks_on_one_gpu(tmp_path):
    """World size EIGHT -- the driver's node -- in the same test mode: the sharded significance pass (seven peers per all_to_all round, 7 views
    per rank) gives the 1-GPU counts, ordered scores and mask; the data-parallel step leaves the same gradient bits on all eight ranks."""
    c = _bench_ranks_on_one_gpu(8, "--mode", "count", "--steps", "7", "--warmup", "1", "--views", "56")["significance_pass"]
    assert c["views"] == 56 and c["rccl_world_size"] == 8 and c["mask_identical_on_all_ranks"] is True
    assert c["mask_equals_1gpu"] is True and c["counts_equal_1gpu"] is True and c["scores_bit_identical_1gpu"] is True and c["hits"] > 0
    d = _bench_ranks_on_one_gpu(8, "--steps", "2", "--warmup", "1", "--views", "8", "--no-c4-leg")
    assert d["n_gpus"] == 8 and d["gradients_identical_on_all_ranks"] is True and d["data_parallel"]["rccl_world_size"] == 8


def test_bench_distill_step_with_two_and_eight_ranks_on_one_gpu(tmp_path):
    """C5's step (student forward + backward at SH degree D - 1 against the teacher's forward, `bench.py --mode distill --gpus W`) in the same
    test mode: the student's gradients after the exchange are the same bits on every rank; at two ranks the rank-one SH exchange and the
    dense all-reduce leave the same bits."""
    a = _bench_ranks_on_one_gpu(2, "--mode", "distill", "--steps", "2", "--warmup", "1", "--views", "8")
    b = _bench_ranks_on_one_gpu(2, "--mode", "distill", "--steps", "2", "--warmup", "1", "--views", "8", "--dense-allreduce")
    assert a["gradients_identical_on_all_ranks"] is True and b["gradients_identical_on_all_ranks"] is True
    assert a["gradient_sha256"] == b["gradient_sha256"], (a["gradient_sha256_per_tensor"], b["gradient_sha256_per_tensor"])
    assert a["data_parallel"]["bytes_on_wire_per_step"] < b["data_parallel"]["bytes_on_wire_per_step"]
    c = _bench_ranks_on_one_gpu(8, "--mode", "distill", "--steps", "2", "--warmup", "1", "--views", "8")
    assert c["n_gpus"] == 8 and c["gradients_identical_on_all_ranks"] is True
