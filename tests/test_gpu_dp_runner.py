"""-m gpu: the data-parallel runner end to end on the device -- `python -m torch.distributed.run --nproc-per-node 1 -m lightgaussian_amd.run
--distributed trainer.py -m out` (RCCL, world size 1, LG_DP_FORCE=1 so that the gradient exchange really goes through the collectives) on a
reference-SHAPED checkout: modules of the reference's names holding its literal formulations (gaussian_renderer.render over the
diff_gaussian_rasterization shim, utils.loss_utils in torch ops, a GaussianModel with the reference's getters and training_setup, a Scene) and
a trainer with the loop of prune_finetune.py:141-168,287-289.  The runner rebinds render / l1_loss / ssim to this package, shards the cameras,
wraps the optimizer; three Adam iterations later the parameters must equal, bit for bit, the same loop run in THIS process on the package's
own functions (world size 1: the average of one rank is the gradient itself), and the exchange must have taken the visible-rows path.
(The reference tree itself is not on the GPU box; its call contracts are pinned on CPU by tests/test_dropin_runner.py.)"""
import json
import math
import os
import socket
import subprocess
import sys

import pytest
import torch

import common
from common import syn

pytestmark = pytest.mark.gpu
ROOT = common.ROOT
N, W, H, NCAM, STEPS = 6000, 160, 96, 6, 3

_RENDER = '''
import math, torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):   # gaussian_renderer/__init__.py:22-124
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device="cuda") + 0
    try: screenspace_points.retain_grad()
    except Exception: pass
    rs = GaussianRasterizationSettings(image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug, f_count=False)
    img, radii = GaussianRasterizer(raster_settings=rs)(means3D=pc.get_xyz, means2D=screenspace_points, shs=pc.get_features, colors_precomp=None,
        opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None)
    return {"render": img, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
def count_render(*a, **k):
    raise NotImplementedError
'''
_LOSS = '''
import torch
def l1_loss(network_output, gt):
    return torch.abs((network_output - gt)).mean()
def ssim(img1, img2, window_size=11, size_average=True):
    raise RuntimeError("the literal ssim must have been rebound by the runner")
'''
_MODEL = f'''
import math, torch
from lightgaussian_amd import synthetic as syn
class GaussianModel(syn.SyntheticGaussians):
    def training_setup(self, training_args):
        ps = [(self._xyz, "xyz", 1e-4), (self._features_dc, "f_dc", 2.5e-3), (self._features_rest, "f_rest", 1.25e-4), (self._opacity, "opacity", 5e-2),
              (self._scaling, "scaling", 5e-3), (self._rotation, "rotation", 1e-3)]
        self.optimizer = torch.optim.Adam([{{"params": [p], "lr": lr, "name": n}} for p, n, lr in ps], lr=0.0, eps=1e-15)
    def prune_points(self, mask):                      # scene/gaussian_model.py:584-600 (not reached by this trainer)
        raise NotImplementedError
    def prune_gaussians(self, percent, import_score):  # scene/gaussian_model.py:776-782
        raise NotImplementedError
def make_model():
    g = syn.make_gaussians({N}, seed=5, log_scale_mean=math.log(0.03)).to("cuda")
    m = GaussianModel(*[torch.nn.Parameter(getattr(g, n).contiguous()) for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")], 3, 3)
    return m
'''
_SCENE = f'''
import torch
from lightgaussian_amd import synthetic as syn
from scene.gaussian_model import GaussianModel, make_model      # scene/__init__.py:12-21 imports the model the same way
class Scene:
    def __init__(self):
        self.train_cameras = {{1.0: [syn.orbit_camera(k, {NCAM}, {W}, {H}).to("cuda") for k in range({NCAM})]}}
        for k, c in enumerate(self.train_cameras[1.0]): c.uid = k
    def getTrainCameras(self, scale=1.0):
        return self.train_cameras[scale]
'''
_TRAINER = f'''
import json, os, sys, random
from random import randint
import torch
from utils.loss_utils import l1_loss, ssim
from gaussian_renderer import render
from scene import Scene, make_model
from lightgaussian_amd import synthetic as syn
random.seed(0); torch.manual_seed(0)                       # utils/general_utils.py:147-150 safe_state
gaussians = make_model(); scene = Scene(); gaussians.training_setup(None)
pipe, background = syn.PipelineParams(), torch.zeros(3, device="cuda")
gts = {{c.uid: torch.rand(3, {H}, {W}, generator=torch.Generator().manual_seed(100 + c.uid)).cuda() for c in scene._lg_all_train_cameras()}} if hasattr(scene, "_lg_all_train_cameras") else None
viewpoint_stack, picked = None, []
iter_start, iter_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # prune_finetune.py:90-91
ema_loss_for_log, timings, kinds = 0.0, [], set()
for iteration in range(1, {STEPS} + 1):                     # prune_finetune.py:133-172,206,287-289
    iter_start.record()
    if not viewpoint_stack:
        viewpoint_stack = scene.getTrainCameras().copy()
    viewpoint_cam = viewpoint_stack.pop(randint(0, len(viewpoint_stack) - 1))
    picked.append(viewpoint_cam.uid)
    render_pkg = render(viewpoint_cam, gaussians, pipe, background)
    image = render_pkg["render"]
    gt_image = gts[viewpoint_cam.uid]
    Ll1 = l1_loss(image, gt_image)
    loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - ssim(image, gt_image))
    loss.backward()
    iter_end.record()
    kinds.add(type(loss).__name__)
    with torch.no_grad():
        ema_loss_for_log = 0.4 * loss.item() + 0.6 * ema_loss_for_log          # prune_finetune.py:172
        timings.append(iter_start.elapsed_time(iter_end))                      # :206, an argument of training_report
        assert Ll1.item() >= 0.0                                               # training_report: tb_writer.add_scalar(..., Ll1.item(), ...)
        gaussians.optimizer.step()
        gaussians.optimizer.zero_grad(set_to_none=True)
from lightgaussian_amd import dp
import torch.distributed as dist
out = os.environ["LG_TEST_OUT"]
_rk = os.environ.get("RANK", "0")
_state = {{n: getattr(gaussians, n).detach().cpu() for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")}}
torch.save(_state, os.path.join(out, f"params_r{{_rk}}.pt"))
if _rk != "0":
    json.dump(dict(picked=picked, stats=dp.stats()), open(os.path.join(out, f"rec_r{{_rk}}.json"), "w"))
    sys.exit(0)
torch.save(_state, os.path.join(out, "params.pt"))
json.dump(dict(picked=picked, stats=dp.stats(), world=dist.get_world_size() if dist.is_initialized() else 0, backend=dist.get_backend() if dist.is_initialized() else None,
               render=render.__module__, wrapped=hasattr(render, "__wrapped__"), argv=sys.argv[1:], ncams=len(scene.getTrainCameras()),
               ema=ema_loss_for_log, timings=[t if t == t else None for t in timings], kinds=sorted(kinds)), open(os.path.join(out, "rec.json"), "w"))
'''


@pytest.mark.parametrize("flags", [[], ["--no-iter-timing"]], ids=["eager-loss", "lazy-loss"])
def test_three_iterations_through_the_distributed_runner_equal_the_in_process_loop(tmp_path, flags):
    """flags = --no-iter-timing (implies --lazy-loss): the trainer's loss line runs on lazy scalars, its per-iteration loss.item() reads
    the pinned copy, its iter_start.elapsed_time(iter_end) does not wait -- and the parameters after three Adam steps are the same bits."""
    root = tmp_path / "LightGaussian"
    for d in ("gaussian_renderer", "utils", "scene"):
        (root / d).mkdir(parents=True)
    (root / "gaussian_renderer" / "__init__.py").write_text(_RENDER)
    (root / "utils" / "__init__.py").write_text("")
    (root / "utils" / "loss_utils.py").write_text(_LOSS)
    (root / "scene" / "gaussian_model.py").write_text(_MODEL)
    (root / "scene" / "__init__.py").write_text(_SCENE)
    (root / "prune.py").write_text("def prune_list(gaussians, scene, pipe, background):\n    raise NotImplementedError\ndef calculate_v_imp_score(gaussians, imp_list, v_pow):\n    raise NotImplementedError\n")
    (root / "trainer.py").write_text(_TRAINER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LG_TEST_OUT=str(tmp_path), LG_DP_FORCE="1", LG_DP_CHECK="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        "-m", "lightgaussian_amd.run", "--distributed"] + flags + [str(root / "trainer.py"), "-m", "out"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, "\n".join(l for l in r.stderr.splitlines() if "rank0" in l or "Error" in l or "error" in l)[-3000:]
    rec = json.load(open(tmp_path / "rec.json"))
    assert rec["world"] == 1 and rec["backend"] == "nccl" and rec["render"] == "lightgaussian_amd.gaussian_renderer" and rec["wrapped"]
    assert rec["argv"] == ["-m", "out"] and rec["ncams"] == NCAM
    assert rec["kinds"] == (["LazyLoss"] if flags else ["Tensor"]) and 0.0 < rec["ema"] < 1.0 and len(rec["timings"]) == STEPS
    if not flags:
        assert all(t is not None and t > 0.0 for t in rec["timings"])       # eager: loss.item() has drained the iteration, the pair is timed
    st = rec["stats"]
    # round 5: the SH groups through the rank-one exchange (all-gather of dRGB behind K9 + lg_sh_grad_from_rgb), the other four tensors
    # through the dense bucketed all-reduce -- both through RCCL
    assert st["steps"] == STEPS and st["rank1_sh_steps"] == STEPS and st["dense_steps"] == STEPS and st["rows_exchanged"] == 0
    assert st["sh_bytes_on_wire"] == STEPS * (3 * N + 3) * 4
    got = torch.load(tmp_path / "params.pt")
    # the same loop in this process on the package's own functions
    import random
    from lightgaussian_amd import loss_utils
    from lightgaussian_amd.gaussian_renderer import render
    dev = torch.device("cuda:0")
    random.seed(0); torch.manual_seed(0)
    g = syn.make_gaussians(N, seed=5, log_scale_mean=math.log(0.03)).to(dev)
    m = syn.SyntheticGaussians(*[torch.nn.Parameter(getattr(g, n).contiguous()) for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")], 3, 3)
    ps = [(m._xyz, 1e-4), (m._features_dc, 2.5e-3), (m._features_rest, 1.25e-4), (m._opacity, 5e-2), (m._scaling, 5e-3), (m._rotation, 1e-3)]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in ps], lr=0.0, eps=1e-15)
    cams = [syn.orbit_camera(k, NCAM, W, H).to(dev) for k in range(NCAM)]
    pipe, bg = syn.PipelineParams(), torch.zeros(3, device=dev)
    stack = None
    for it in range(STEPS):
        if not stack:
            stack = list(range(NCAM))
        k = stack.pop(random.randint(0, len(stack) - 1))
        assert k == rec["picked"][it]
        gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(100 + k)).to(dev)
        image = render(cams[k], m, pipe, bg)["render"]
        loss = 0.8 * loss_utils.l1_loss(image, gt) + 0.2 * (1.0 - loss_utils.ssim(image, gt))
        loss.backward()
        with torch.no_grad():
            opt.step(); opt.zero_grad(set_to_none=True)
    for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(got[n], getattr(m, n).detach().cpu()), n


def _write_checkout(tmp_path):
    root = tmp_path / "LightGaussian"
    for d in ("gaussian_renderer", "utils", "scene"):
        (root / d).mkdir(parents=True)
    (root / "gaussian_renderer" / "__init__.py").write_text(_RENDER)
    (root / "utils" / "__init__.py").write_text("")
    (root / "utils" / "loss_utils.py").write_text(_LOSS)
    (root / "scene" / "gaussian_model.py").write_text(_MODEL)
    (root / "scene" / "__init__.py").write_text(_SCENE)
    (root / "prune.py").write_text("def prune_list(gaussians, scene, pipe, background):\n    raise NotImplementedError\ndef calculate_v_imp_score(gaussians, imp_list, v_pow):\n    raise NotImplementedError\n")
    (root / "trainer.py").write_text(_TRAINER)
    return root


def test_two_ranks_of_the_distributed_runner_on_one_gpu_train_one_model(tmp_path):
    """`run.py --distributed` at world size 2 with the real kernels: both ranks on the one GPU of this box, collectives over gloo on device
    tensors (--backend=gloo).  The reference-shaped trainer (camera shard per rank, loss line, loss.item(), Adam) runs three iterations:
    both ranks end with the same parameters, bit for bit, in the round-5 exchange (rank-one SH gradients + dense rest) and in the dense
    one (LG_DP_SH=dense), and the two exchanges leave the same parameters."""
    res = {}
    for mode in ("rank1", "dense"):
        out = tmp_path / mode
        out.mkdir()
        root = _write_checkout(out)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, LG_TEST_OUT=str(out), LG_DP_SH=mode, LG_DP_CHECK_SET="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                            "-m", "lightgaussian_amd.run", "--distributed", "--backend=gloo", str(root / "trainer.py"), "-m", "out"],
                           capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
        assert r.returncode == 0, "\n".join(l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "rank" in l)[-3000:]
        p0, p1 = torch.load(out / "params_r0.pt"), torch.load(out / "params_r1.pt")
        for n in p0:
            assert torch.equal(p0[n], p1[n]), f"{mode}: {n} differs between the ranks"
        rec0, rec1 = json.load(open(out / "rec.json")), json.load(open(out / "rec_r1.json"))
        assert rec0["world"] == 2 and rec0["backend"] == "gloo" and not (set(rec0["picked"]) & set(rec1["picked"]))     # disjoint camera shards
        assert rec0["stats"]["rank1_sh_steps"] == (STEPS if mode == "rank1" else 0)
        res[mode] = p0
    for n in res["rank1"]:
        assert torch.equal(res["rank1"][n], res["dense"][n]), f"{n}: the rank-one exchange and the dense exchange trained different models"
    g = syn.make_gaussians(N, seed=5, log_scale_mean=math.log(0.03))
    assert not torch.equal(res["rank1"]["_features_rest"], g._features_rest)        # and they did train
