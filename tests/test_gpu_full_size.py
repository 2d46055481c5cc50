"""-m gpu: oracle parity AT BASELINE.json's sizes (r2 verdict: C5 had only a timing, C2 only invariants, and the full-size
gradient parity was a line bench.py printed -- nothing that could fail).

  C2  1 M Gaussians, 1920x1080, forward           image bit-identical to the float32 oracle in canonical mode, n_contrib equal
  C3  3 M Gaussians, 1920x1080, forward + backward image <= 1e-4, every raw-parameter gradient <= 1e-4 (tensor level AND element-wise)
  C5  6 M Gaussians, 1600x1060, distillation shape  teacher (SH degree 3) forward + student (degree 2, M = 9) forward + backward
                                                    (distill_train.py:124-146: the per-iteration body)
  C3L 3 M Gaussians of 3x the size (sigma 0.012: 11 M tile instances, lists of ~1400 entries -- BASELINE configs[2] says
      "MipNeRF360 'room'-scale"; the frozen generator's 4 M instances are light next to a capture) and the heavy-tailed scene (a pile
      of 60 k faint splats every camera looks at: lists of 20-25 k entries, 40+ segments, the long-tile kernels of forward and backward):
      the same checks as C3 plus the significance outputs -- counts, scores (default weight AND alpha T) bit-identical   (round 6, r5 verdict)

How the comparison is set up.  The product renders through gaussian_renderer.render() with the getters fused into the kernels
(the raw GaussianModel tensors go in; gradients come back on them).  The oracle (oracle/lg_oracle.c, OpenMP on the host cores)
takes ACTIVATED inputs; to feed it exactly what the kernels see, the activations are evaluated by torch ON THE DEVICE (the literal
getter pattern, bit-identical to the fused evaluation: tests/test_gpu_parity.py::test_fused_getters_match_unfused_render) and
copied to the host.  The oracle's gradients (w.r.t. the activated tensors, float64) are chained to the raw parameters through the
reference's getters (scene/gaussian_model.py:98-118: exp / sigmoid / F.normalize / cat) by float64 torch autograd on the CPU.

Tolerances (BASELINE.json north_star: "rendered RGB and gradients within 1e-4 rel"):
  image       max |a - b| <= 1e-4 max|b|   (observed ~4e-7); canonical mode: bit-identical
  gradients   vs the float32 oracle (the restatement of the reference's float32 CUDA path):
              tensor level  max |a - b| <= 1e-4 max |b|
              element-wise  |a - b| <= 1e-4 |b| + 2e-5 max|b| on the WELL-CONDITIONED entries -- those on which the float32 oracle
              itself stays within a quarter of that bound of the float64 one -- and, on the rest, within 3x the float32 oracle's own
              distance from the float64 one (the published back-to-front replay T / (1 - alpha) has a float32 noise floor, and
              float64 takes a handful of threshold decisions the other way)
  n_contrib   compared as the Gaussian id of each pixel's last contributor (the index itself is a position in the library's culled
              tile lists) and final T.  Canonical mode: equal bit for bit; hardware-exp mode: ids equal except on pixels whose
              transmittance crosses 1e-4 within rounding (<= 1e-4 of the pixels)
"""
import math

import numpy as np
import pytest
import torch

import common
import gpu_common
from common import syn
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _activated_on_device(pc):
    """The reference's getters, evaluated by torch on the device (what render()'s literal pattern feeds the rasterizer)."""
    with torch.no_grad():
        return dict(means3D=pc.get_xyz, opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation,
                    shs=pc.get_features.contiguous())


def _oracle_kw(act, cam, W, H, deg, bg):
    kw = {k: v.detach().cpu().numpy() for k, v in act.items()}
    kw.update(W=W, H=H, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=np.asarray(bg, np.float32),
              viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
              campos=cam.camera_center.cpu().numpy(), sh_degree=deg)
    return kw


def _last_contributor(image, W, H):
    """Gaussian id of every pixel's last contributor (0xFFFFFFFF: none) and the final transmittance, from the state the forward
    saved for its backward.  (n_contrib itself is a position in the library's CULLED tile lists; the oracle enumerates the
    reference's full rectangles -- the id is what the two share.)"""
    import ctypes as C
    from lightgaussian_amd import _lib, rasterizer
    fn = image.grad_fn
    *_, radii, geom, binning, img = fn.saved_tensors
    call = rasterizer._Call(fn.raster_settings, fn.saved_tensors[0], None, None, None, None, None, None, exact=False, opts=fn.opts)
    out = torch.empty(W * H, dtype=torch.int32, device=image.device)
    _lib.check(_lib.load().lg_debug_last_contributor(C.byref(call.view), radii.shape[0], geom.data_ptr(), binning.data_ptr(), img.data_ptr(),
                                                     int(fn.num_rendered), out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out.cpu().numpy().view(np.uint32), img[: W * H * 4].view(torch.float32).cpu().numpy()


def _chain_to_raw(g_cpu, grads_act):
    """Oracle gradients w.r.t. the activated tensors -> gradients w.r.t. the raw parameters, through the reference's getters in
    float64 torch autograd on the CPU."""
    raw = {n: getattr(g_cpu, n).detach().double().requires_grad_(True) for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")}
    outs = [raw["_xyz"] * 1.0, torch.sigmoid(raw["_opacity"]), torch.exp(raw["_scaling"]), torch.nn.functional.normalize(raw["_rotation"]),
            torch.cat((raw["_features_dc"], raw["_features_rest"]), dim=1)]
    gs = [torch.from_numpy(np.asarray(grads_act[k], np.float64)).reshape(o.shape) for k, o in zip(("means3D", "opacities", "scales", "rotations", "shs"), outs)]
    torch.autograd.backward(outs, gs)
    return {n: t.grad.numpy() for n, t in raw.items()}


def _check_grads(hip_raw, ref64_raw, ref32_raw):
    """hip vs the float32 oracle (the restatement of the reference's float32 CUDA path: THE parity contract) at tensor level, and
    element-wise wherever float32 itself is trustworthy -- judged by the float32 oracle's own distance from the float64 one."""
    report = {}
    for name, r64 in ref64_raw.items():
        r32 = ref32_raw[name].reshape(r64.shape)
        a = hip_raw[name].reshape(r64.shape).astype(np.float64)
        scale = np.abs(r32).max() + 1e-300
        tensor_err = np.abs(a - r32).max() / scale
        bound = 1e-4 * np.abs(r32) + 2e-5 * scale
        noise32 = np.abs(r32 - r64)                         # what float32 arithmetic (rounding, and the rare threshold decision that
        well = noise32 <= 0.25 * bound                      #   falls the other way in float64) does to this entry
        ratio = np.abs(a - r32) / bound
        worst_well = float(ratio[well].max()) if well.any() else 0.0
        # elsewhere: within 3x of the float32 oracle's own distance from the float64 one (or the bound, whichever is larger)
        loose = np.abs(a - r32) <= np.maximum(bound, 3.0 * noise32)
        report[name] = dict(tensor_rel_err=float(tensor_err), worst_well_conditioned_over_bound=worst_well, well_fraction=float(well.mean()),
                            outside_3x_noise=int((~loose).sum()), entries=int(a.size))
        assert tensor_err <= 1e-4, f"{name}: tensor-level rel err {tensor_err:.3e} vs the float32 oracle"
        assert well.mean() > 0.99, f"{name}: only {well.mean():.4f} of the entries are well-conditioned in float32"
        assert worst_well <= 1.0, f"{name}: element-wise bound missed on a well-conditioned entry ({worst_well:.2f}x)"
        assert (~loose).mean() <= 1e-6, f"{name}: {int((~loose).sum())} entries further from the float32 oracle than 3x its own float32 noise"
    return report


def _fwd_bwd_case(N, W, H, max_deg, view, seed_img, scale=None, heavy=False):
    """render() -> sum(image * g) -> backward on the raw parameters, against the oracle on the same activated inputs."""
    from lightgaussian_amd.gaussian_renderer import render
    from lightgaussian_amd import parallel
    dev = torch.device(DEV)
    # the frozen SURVEY 8d scene (SH degree 3 storage); scale: another median sigma (bench.py --scale); heavy: bench.py --scene heavy
    g3 = syn.make_gaussians(N) if scale is None else syn.make_gaussians(N, log_scale_mean=math.log(scale))
    if heavy:
        syn.make_heavy_tailed(g3)
    g_cpu = g3 if max_deg == 3 else parallel.make_student(g3, max_deg)     # distill_train.py:78-79 + onedownSHdegree
    pc = g_cpu.to(dev).requires_grad_(True)
    cam = syn.orbit_camera(view, 200, W, H)
    camd = cam.to(dev)
    bg = torch.zeros(3, device=dev)
    pipe = syn.PipelineParams()
    gimg = np.random.RandomState(seed_img).randn(3, H, W).astype(np.float32) / (3 * H * W)
    image = render(camd, pc, pipe, bg)["render"]
    last_ids, _final_T = _last_contributor(image, W, H)
    (image * torch.from_numpy(gimg).to(dev)).sum().backward()
    hip_raw = {n: getattr(pc, n).grad.detach().cpu().numpy() for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")}
    hip_img = image.detach().cpu().numpy()
    kw = _oracle_kw(_activated_on_device(pc), cam, W, H, max_deg, np.zeros(3))
    del image
    f32 = oracle.forward(**kw); g32 = oracle.backward(f32, gimg)
    f64 = oracle.forward(dtype=np.float64, **kw); g64 = oracle.backward(f64, gimg)
    # forward
    assert gpu_common.rel_err(hip_img, f32.color) <= 1e-4
    # (no per-pixel bound against the float64 oracle: it takes some alpha >= 1/255 / radius decisions the other way)
    ids_ref = oracle.last_contributor_ids(f32)
    assert np.mean(last_ids != ids_ref) <= 1e-4, f"last contributor differs on {np.mean(last_ids != ids_ref):.2e} of the pixels"
    # backward, raw parameters
    rep = _check_grads(hip_raw, _chain_to_raw(g_cpu, g64), _chain_to_raw(g_cpu, g32))
    vis = int((f32.radii > 0).sum())
    return rep, vis, f32.num_rendered, g_cpu, pc, camd, kw, f32


def test_c3_full_size_image_and_every_raw_parameter_gradient_match_the_oracle():
    """BASELINE configs[2]: 3 M Gaussians, 1080p, SH degree 3, fwd + bwd through render() (the step bench.py times)."""
    rep, vis, R, *_ = _fwd_bwd_case(3_000_000, 1920, 1080, 3, view=0, seed_img=0)
    assert vis > 1_500_000 and R > 3_000_000
    print("C3 gradient parity (tensor rel err, worst well-conditioned / bound, worst / bound, fp32 oracle worst / bound, well fraction):", rep)


def test_c5_distillation_shape_teacher_forward_and_student_forward_backward_match_the_oracle():
    """BASELINE configs[4] / distill_train.py:124-146 at its own shape: 6 M Gaussians, 1600 x 1060; the student at SH degree 2
    (M = 9: _features_rest [N, 8, 3]) is differentiated, the teacher at degree 3 is rendered forward only."""
    from lightgaussian_amd.gaussian_renderer import render
    from lightgaussian_amd import rasterizer
    N, W, H = 6_000_000, 1600, 1060
    rep, vis, R, g_student, pc, camd, kw_s, f32_s = _fwd_bwd_case(N, W, H, 2, view=37, seed_img=1)
    assert g_student._features_rest.shape == (N, 8, 3) and g_student.active_sh_degree == 2
    assert vis > 3_000_000 and R > 6_000_000
    print("C5 student gradient parity:", rep)
    dev = torch.device(DEV)
    bg, pipe = torch.zeros(3, device=dev), syn.PipelineParams()
    # student, canonical arithmetic: image bit-identical, n_contrib equal
    pcs = g_student.to(dev).requires_grad_(True)
    img = render(camd, pcs, pipe, bg, options={"fast_exp": False})["render"]
    assert np.array_equal(img.detach().cpu().numpy().view(np.uint32), f32_s.color.view(np.uint32))
    ids, fT = _last_contributor(img, W, H)
    assert np.array_equal(ids, oracle.last_contributor_ids(f32_s)) and np.array_equal(fT.view(np.uint32), f32_s.saved["final_T"].view(np.uint32))
    del img, pcs, pc, f32_s
    torch.cuda.empty_cache()
    # teacher: SH degree 3 forward (no grad), hardware exp as distill_step renders it, and canonical
    teacher = syn.make_gaussians(N).to(dev)
    cam = syn.orbit_camera(37, 200, W, H)
    kw_t = _oracle_kw(_activated_on_device(teacher), cam, W, H, 3, np.zeros(3))
    ref = oracle.forward(**kw_t)
    with torch.no_grad():
        fast = render(camd, teacher, pipe, bg)
        exact = render(camd, teacher, pipe, bg, options={"fast_exp": False})
    assert np.array_equal(fast["radii"].cpu().numpy(), ref.radii) and np.array_equal(exact["radii"].cpu().numpy(), ref.radii)
    assert gpu_common.rel_err(fast["render"].cpu().numpy(), ref.color) <= 1e-4
    assert np.array_equal(exact["render"].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))
    assert rasterizer.resolve_options()["fast_exp"] is True            # per-call option: the default was not touched


def test_c2_one_million_gaussians_1080p_forward_is_bit_identical_to_the_oracle():
    """BASELINE configs[1]: 1 M Gaussians, 1080p, forward only.  Canonical mode: image, radii, n_contrib and the significance
    outputs equal the float32 oracle bit for bit; the default (hardware-exp) render stays within 1e-4."""
    from lightgaussian_amd.gaussian_renderer import render, count_render
    dev = torch.device(DEV)
    N, W, H = 1_000_000, 1920, 1080
    pc = syn.make_gaussians(N).to(dev)
    cam = syn.orbit_camera(11, 200, W, H)
    camd, bg, pipe = cam.to(dev), torch.zeros(3, device=dev), syn.PipelineParams()
    ref = oracle.forward(count=True, **_oracle_kw(_activated_on_device(pc), cam, W, H, 3, np.zeros(3)))
    pcg = syn.make_gaussians(N).to(dev).requires_grad_(True)         # (grad-enabled so that the per-pixel state is reachable)
    exact = render(camd, pcg, pipe, bg, options={"fast_exp": False})
    assert np.array_equal(exact["radii"].cpu().numpy(), ref.radii)
    assert np.array_equal(exact["render"].detach().cpu().numpy().view(np.uint32), ref.color.view(np.uint32))
    ids, fT = _last_contributor(exact["render"], W, H)
    assert np.array_equal(ids, oracle.last_contributor_ids(ref)) and np.array_equal(fT.view(np.uint32), ref.saved["final_T"].view(np.uint32))
    with torch.no_grad():
        fast = render(camd, pc, pipe, bg)["render"].cpu().numpy()
        cnt = count_render(camd, pc, pipe, bg)
    assert gpu_common.rel_err(fast, ref.color) <= 1e-4
    assert np.array_equal(cnt["gaussians_count"].cpu().numpy(), ref.count)
    assert np.array_equal(cnt["important_score"].cpu().numpy().view(np.uint32), ref.score.view(np.uint32))
    assert np.array_equal(cnt["render"].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))


def test_a_key_that_really_exceeds_64_bits_renders_like_the_oracle():
    """tile | depth | id beyond 64 bits without the cross-check switch: 1.1 M Gaussians (21 id bits) on 4112 x 4096 pixels (257 x 256
    = 65 792 tiles: 17 bits).  The exact forward lays the depth out from the view's own maximum (26 bits here: 64 in all, it fits);
    the capacity-bounded forward -- the default from the second view of a shape on -- from the camera's zfar (27 bits: 65), so the
    lowest depth bit is left out of the stored key and lg_tile_ranges completes the order from the binning record.  Both must
    equal the oracle bit for bit (r2: this shape fell back to a hipCUB pair sort and had no bounded form at all)."""
    from lightgaussian_amd import rasterizer
    from lightgaussian_amd.gaussian_renderer import count_render
    dev = torch.device(DEV)
    N, W, H = 1_100_000, 4112, 4096
    pc = syn.make_gaussians(N, log_scale_mean=math.log(0.006)).to(dev)
    cam = syn.orbit_camera(5, 200, W, H)
    camd, bg, pipe = cam.to(dev), torch.zeros(3, device=dev), syn.PipelineParams()
    ref = oracle.forward(count=True, **_oracle_kw(_activated_on_device(pc), cam, W, H, 3, np.zeros(3)))
    key = (dev.index, N, W, H)
    with rasterizer._CAP_LOCK:
        rasterizer._CAPACITY.pop(key, None)
    with torch.no_grad(), rasterizer.options(sync_free="validated"):
        first = count_render(camd, pc, pipe, bg)            # exact forward (learns the capacity of this shape)
        assert rasterizer._CAPACITY.get(key, 0) > 0
        second = count_render(camd, pc, pipe, bg)           # bounded forward: 17 + 27 + 21 = 65 key bits
    for out in (first, second):
        assert np.array_equal(out["radii"].cpu().numpy(), ref.radii)
        assert np.array_equal(out["gaussians_count"].cpu().numpy(), ref.count)
        assert np.array_equal(out["important_score"].cpu().numpy().view(np.uint32), ref.score.view(np.uint32))
        assert np.array_equal(out["render"].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))
    assert int(ref.count.sum()) > 10_000_000


def _count_case(pc, cam, camd, ref, W, H):
    """count_render (image-returning, canonical) and the significance-only pass's forward against the oracle: everything bit for bit."""
    from lightgaussian_amd.gaussian_renderer import count_render
    bg, pipe = torch.zeros(3, device=torch.device(DEV)), syn.PipelineParams()
    with torch.no_grad():
        full = count_render(camd, pc, pipe, bg)
        sig = count_render(camd, pc, pipe, bg, options={"skip_color_in_count": True})
    for out in (full, sig):
        assert np.array_equal(out["radii"].cpu().numpy(), ref.radii)
        assert np.array_equal(out["gaussians_count"].cpu().numpy(), ref.count)
        assert np.array_equal(out["important_score"].cpu().numpy().view(np.uint32), ref.score.view(np.uint32))
    assert np.array_equal(full["render"].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))


@pytest.mark.parametrize("kind", ["large_splats", "heavy_tailed"])
def test_c3_size_long_lists_forward_backward_and_significance_match_the_oracle(kind):
    """BASELINE configs[2] at list lengths closer to a capture's (r5 verdict: full-size parity existed on the small-splat scene only, the
    heavier scenes of bench.py carried rates without a check): image (hardware exp <= 1e-4, canonical bit-identical), last contributors,
    every raw-parameter gradient through _check_grads, hit counts and scores bit-identical."""
    from lightgaussian_amd.gaussian_renderer import render
    N, W, H = 3_000_000, 1920, 1080
    kw = dict(scale=0.012) if kind == "large_splats" else dict(heavy=True)
    rep, vis, R, g_cpu, pc, camd, okw, f32 = _fwd_bwd_case(N, W, H, 3, view=23, seed_img=2, **kw)
    print(f"C3L {kind}: {vis} visible, {R} instances; gradient parity:", rep)
    if kind == "large_splats":
        assert R > 9_000_000
    else:
        assert R > 4_500_000
    dev = torch.device(DEV)
    bg, pipe = torch.zeros(3, device=dev), syn.PipelineParams()
    pcs = g_cpu.to(dev).requires_grad_(True)
    img = render(camd, pcs, pipe, bg, options={"fast_exp": False})["render"]
    assert np.array_equal(img.detach().cpu().numpy().view(np.uint32), f32.color.view(np.uint32))
    ids, fT = _last_contributor(img, W, H)
    assert np.array_equal(ids, oracle.last_contributor_ids(f32)) and np.array_equal(fT.view(np.uint32), f32.saved["final_T"].view(np.uint32))
    del img, pcs, pc, f32
    torch.cuda.empty_cache()
    cam = syn.orbit_camera(23, 200, W, H)
    model = g_cpu.to(dev)
    ref = oracle.forward(count=True, **okw)
    _count_case(model, cam, camd, ref, W, H)


def test_c3_size_alpha_t_weights_are_bit_identical_to_the_oracle():
    """The per-hit weight policy at BASELINE configs[2]'s size (3 M Gaussians, 1080p, ~110 M hits): counts and Q24.40 scores of the
    significance-only pass equal the oracle's bit for bit, two runs equal each other (no float atomics), and alpha differs from alpha T."""
    from lightgaussian_amd.gaussian_renderer import count_render
    dev = torch.device(DEV)
    N, W, H = 3_000_000, 1920, 1080
    pc = syn.make_gaussians(N).to(dev)
    cam = syn.orbit_camera(5, 200, W, H)
    camd, bg, pipe = cam.to(dev), torch.zeros(3, device=dev), syn.PipelineParams()
    okw = _oracle_kw(_activated_on_device(pc), cam, W, H, 3, np.zeros(3))
    outs = {}
    for pol, opol in (("alpha_t", oracle.W_ALPHA_T), ("alpha", oracle.W_ALPHA)):
        ref = oracle.forward(count=True, weight_policy=opol, **okw)
        with torch.no_grad():
            a = count_render(camd, pc, pipe, bg, options={"skip_color_in_count": True, "weight_policy": pol})
            b = count_render(camd, pc, pipe, bg, options={"skip_color_in_count": True, "weight_policy": pol})
        for out in (a, b):
            assert np.array_equal(out["gaussians_count"].cpu().numpy(), ref.count)
            bad = np.flatnonzero(out["important_score"].cpu().numpy().view(np.uint32) != ref.score.view(np.uint32))
            assert bad.size == 0, f"{pol}: {bad.size} scores differ from the oracle (first: Gaussian {bad[0] if bad.size else None})"
        assert int(ref.count.sum()) > 50_000_000
        outs[pol] = ref.score
    assert not np.array_equal(outs["alpha"], outs["alpha_t"])
