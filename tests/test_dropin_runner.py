"""CPU: lightgaussian_amd.run.patch_reference() on the REFERENCE's own modules (imported unmodified over the shims, as
tests/test_dropin_reference_modules.py does): every symbol of the runner's table is rebound, in the owning module AND in a
module that had imported it by name beforehand (what the trainers do), signatures stay call-compatible, and
unpatch_reference() restores the originals.  (The GPU behaviour of the rebound paths: tests/test_gpu_dropin_runner.py.)"""
import importlib
import inspect
import sys
import types

import pytest
import torch

import dropin_common
from lightgaussian_amd import run as lg_run

pytestmark = pytest.mark.skipif(not dropin_common.available(), reason="/root/reference is not present on this box")


@pytest.fixture()
def ref():
    mods = dropin_common.load()
    lu = importlib.import_module("utils.loss_utils")
    vq = importlib.import_module("vectree.vq")
    yield mods + (lu, vq)
    lg_run.unpatch_reference()


def _positional(fn):
    return [p.name for p in inspect.signature(fn).parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]


def test_every_symbol_of_the_table_is_rebound_and_restored(ref):
    gr, gm, pr, lu, vq = ref
    # a "trainer" that imported the names before the patch, the way prune_finetune.py:15-17,39 does
    trainer = types.ModuleType("fake_trainer")
    trainer.render, trainer.count_render = gr.render, gr.count_render
    trainer.l1_loss, trainer.ssim = lu.l1_loss, lu.ssim
    trainer.prune_list, trainer.calculate_v_imp_score = pr.prune_list, pr.calculate_v_imp_score
    trainer.renamed_render = gr.render
    sys.modules["fake_trainer"] = trainer
    originals = {"render": gr.render, "count_render": gr.count_render, "l1_loss": lu.l1_loss, "ssim": lu.ssim, "prune_list": pr.prune_list,
                 "calculate_v_imp_score": pr.calculate_v_imp_score, "prune_points": gm.GaussianModel.prune_points,
                 "prune_gaussians": gm.GaussianModel.prune_gaussians, "gumbel_sample": vq.gumbel_sample, "vq_torch": vq.torch}
    try:
        report = lg_run.patch_reference()
        from lightgaussian_amd import gaussian_renderer as lg_gr, loss_utils as lg_loss
        assert gr.render is lg_gr.render and gr.count_render is lg_gr.count_render
        assert lu.l1_loss is lg_loss.l1_loss and lu.ssim is lg_loss.ssim
        assert pr.prune_list is lg_run._prune_list and pr.calculate_v_imp_score is lg_run._calculate_v_imp_score
        assert gm.GaussianModel.prune_points is lg_run._prune_points and gm.GaussianModel.prune_gaussians is lg_run._prune_gaussians
        assert vq.gumbel_sample is not originals["gumbel_sample"] and vq.torch is not torch and vq.torch.cdist is not torch.cdist
        assert vq.torch.zeros is torch.zeros and vq.torch.nn is torch.nn          # everything else is torch's own
        # the already-imported names followed (also under another name)
        assert trainer.render is lg_gr.render and trainer.renamed_render is lg_gr.render and trainer.count_render is lg_gr.count_render
        assert trainer.l1_loss is lg_loss.l1_loss and trainer.ssim is lg_loss.ssim
        assert trainer.prune_list is lg_run._prune_list and trainer.calculate_v_imp_score is lg_run._calculate_v_imp_score
        for key in ("gaussian_renderer.render", "gaussian_renderer.count_render", "utils.loss_utils.l1_loss", "utils.loss_utils.ssim", "prune.prune_list",
                    "prune.calculate_v_imp_score", "scene.gaussian_model.GaussianModel.prune_points",
                    "scene.gaussian_model.GaussianModel.prune_gaussians", "vectree.vq.gumbel_sample"):
            assert key in report and "skipped" not in report[key], (key, report.get(key))
        assert report["gaussian_renderer.render"]["also_rebound_in"] >= 2
        assert lg_run.patch_reference() == report                                  # idempotent
        # call contracts: the reference's positional parameters, in order, are accepted by the replacements
        for name, new in (("render", gr.render), ("count_render", gr.count_render), ("l1_loss", lu.l1_loss), ("ssim", lu.ssim),
                          ("prune_list", pr.prune_list), ("calculate_v_imp_score", pr.calculate_v_imp_score),
                          ("prune_points", gm.GaussianModel.prune_points), ("prune_gaussians", gm.GaussianModel.prune_gaussians),
                          ("gumbel_sample", vq.gumbel_sample)):
            want, have = _positional(originals[name]), _positional(new)
            assert have[:len(want)] == want, (name, want, have)
            for p_old, p_new in zip(inspect.signature(originals[name]).parameters.values(), inspect.signature(new).parameters.values()):
                if p_old.default is not inspect.Parameter.empty:
                    assert p_new.default == p_old.default, (name, p_old.name)
    finally:
        lg_run.unpatch_reference()
        del sys.modules["fake_trainer"]
    assert gr.render is originals["render"] and gr.count_render is originals["count_render"]
    assert lu.l1_loss is originals["l1_loss"] and lu.ssim is originals["ssim"]
    assert pr.prune_list is originals["prune_list"] and pr.calculate_v_imp_score is originals["calculate_v_imp_score"]
    assert gm.GaussianModel.prune_points is originals["prune_points"] and gm.GaussianModel.prune_gaussians is originals["prune_gaussians"]
    assert vq.gumbel_sample is originals["gumbel_sample"] and vq.torch is torch
    assert trainer.render is originals["render"] and trainer.ssim is originals["ssim"]


def test_patched_vectree_search_leaves_cpu_and_unsupported_calls_to_torch(ref):
    """The deferred-cdist proxy only takes over p = 2 distances of HIP float32 tensors with d <= 63; on CPU tensors the
    reference's EuclideanCodebook.forward and kmeans() run exactly as before (same indices, same quantised rows)."""
    *_, vq = ref
    torch.manual_seed(0)
    cb = vq.EuclideanCodebook(dim=12, codebook_size=64, kmeans_init=False, decay=0.8).eval()
    x = torch.randn(1, 500, 12)
    q0, i0 = cb(x)
    samples = torch.randn(1, 300, 12)
    torch.manual_seed(1); m0, b0 = vq.kmeans(samples, 16, 3)
    lg_run.patch_reference()
    q1, i1 = cb(x)
    torch.manual_seed(1); m1, b1 = vq.kmeans(samples, 16, 3)
    assert torch.equal(i0, i1) and torch.equal(q0, q1) and torch.equal(m0, m1) and torch.equal(b0, b1)
    # the deferred object behaves like the tensor when anything but the argmax is asked of it
    lazy = -lg_run._LazyNegDist(x[0], cb.embed[0])
    assert torch.equal(lazy.materialize(), -torch.cdist(x[0], cb.embed[0], p=2))
    assert lazy.shape == (500, 64)


def test_runner_command_line_sets_up_the_path_like_python_does(tmp_path, monkeypatch):
    script = tmp_path / "trainer.py"
    script.write_text("import sys, json\nimport diff_gaussian_rasterization as d\nprint(json.dumps({'argv': sys.argv[1:], 'p0': sys.path[0], 'shim': d.__file__}))\n")
    import subprocess
    import json
    out = subprocess.run([sys.executable, "-m", "lightgaussian_amd.run", "--no-patch", str(script), "-s", "scene", "--iterations", "3"],
                         capture_output=True, text=True, cwd=dropin_common.ROOT, check=True).stdout.strip().splitlines()[-1]
    rec = json.loads(out)
    assert rec["argv"] == ["-s", "scene", "--iterations", "3"] and rec["p0"] == str(tmp_path)
    assert rec["shim"].startswith(dropin_common.ROOT)
    bad = subprocess.run([sys.executable, "-m", "lightgaussian_amd.run", "--frobnicate", str(script)], capture_output=True, text=True, cwd=dropin_common.ROOT)
    assert bad.returncode != 0 and "unknown option" in bad.stderr
    # --weight-policy=NAME: the process default of every count_render of the run (round 6); an unknown name is refused before the script runs
    script.write_text("import json\nfrom lightgaussian_amd import rasterizer, _lib\nprint(json.dumps({'wp': rasterizer.resolve_options()['weight_policy'], 'alpha_t': _lib.WEIGHT_ALPHA_T}))\n")
    out = subprocess.run([sys.executable, "-m", "lightgaussian_amd.run", "--no-patch", "--weight-policy=alpha_t", str(script)],
                         capture_output=True, text=True, cwd=dropin_common.ROOT, check=True).stdout.strip().splitlines()[-1]
    rec = json.loads(out)
    assert rec["wp"] == rec["alpha_t"]
    bad = subprocess.run([sys.executable, "-m", "lightgaussian_amd.run", "--weight-policy=alphaT", str(script)], capture_output=True, text=True, cwd=dropin_common.ROOT)
    assert bad.returncode != 0 and "weight_policy" in bad.stderr


def test_runner_rebinds_what_the_trainer_script_then_imports_by_name(tmp_path):
    """The whole command-line flow on a reference-SHAPED checkout (a temporary directory with gaussian_renderer/, utils/loss_utils.py,
    prune.py holding stand-ins): `python -m lightgaussian_amd.run trainer.py` patches first and runs the script afterwards, so the
    trainer's own `from gaussian_renderer import render, count_render` / `from utils.loss_utils import l1_loss, ssim` /
    `from prune import prune_list` bind this package's implementations; with --no-patch they bind the checkout's."""
    import json
    import subprocess
    root = tmp_path / "LightGaussian"
    (root / "gaussian_renderer").mkdir(parents=True)
    (root / "utils").mkdir()
    (root / "gaussian_renderer" / "__init__.py").write_text("def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):\n    return 'literal'\n"
                                                            "def count_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):\n    return 'literal'\n")
    (root / "utils" / "__init__.py").write_text("")
    (root / "utils" / "loss_utils.py").write_text("def l1_loss(network_output, gt):\n    return 'literal'\ndef ssim(img1, img2, window_size=11, size_average=True):\n    return 'literal'\n")
    (root / "prune.py").write_text("def prune_list(gaussians, scene, pipe, background):\n    return 'literal'\ndef calculate_v_imp_score(gaussians, imp_list, v_pow):\n    return 'literal'\n")
    (root / "trainer.py").write_text(
        "import json, sys\n"
        "from utils.loss_utils import l1_loss, ssim\n"
        "from gaussian_renderer import render, count_render\n"
        "from prune import prune_list, calculate_v_imp_score\n"
        "print(json.dumps({n: f.__module__ for n, f in dict(render=render, count_render=count_render, l1_loss=l1_loss, ssim=ssim,\n"
        "                  prune_list=prune_list, calculate_v_imp_score=calculate_v_imp_score).items()}))\n")
    def run(*flags):
        out = subprocess.run([sys.executable, "-m", "lightgaussian_amd.run", *flags, str(root / "trainer.py")], capture_output=True, text=True,
                             cwd=dropin_common.ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    patched = run()
    assert patched["render"] == "lightgaussian_amd.gaussian_renderer" and patched["count_render"] == "lightgaussian_amd.gaussian_renderer"
    assert patched["l1_loss"] == "lightgaussian_amd.loss_utils" and patched["ssim"] == "lightgaussian_amd.loss_utils"
    # (the adapters live in run.py, which `python -m` executes under the name __main__)
    assert patched["prune_list"] in ("lightgaussian_amd.run", "__main__") and patched["calculate_v_imp_score"] in ("lightgaussian_amd.run", "__main__")
    literal = run("--no-patch")
    assert literal == {"render": "gaussian_renderer", "count_render": "gaussian_renderer", "l1_loss": "utils.loss_utils", "ssim": "utils.loss_utils",
                       "prune_list": "prune", "calculate_v_imp_score": "prune"}


def test_fused_adam_switch_only_touches_cuda_parameter_lists_and_restores():
    """run.py --fused-adam: torch.optim.Adam as the trainers construct it (scene/gaussian_model.py:training_setup), fused=True added
    for CUDA parameters only; explicit fused / foreach arguments and CPU parameters are left alone; switching off restores torch's
    own constructor."""
    orig = torch.optim.Adam.__init__
    lg_run.fused_adam(True)
    try:
        assert torch.optim.Adam.__init__ is not orig
        p = [torch.nn.Parameter(torch.randn(4, 3))]
        opt = torch.optim.Adam([{"params": p, "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)      # the reference's call shape
        assert not opt.defaults.get("fused")                                                       # CPU parameters: untouched
        p[0].grad = torch.ones_like(p[0])
        opt.step()
        assert torch.optim.Adam([torch.nn.Parameter(torch.randn(2))], foreach=False).defaults["foreach"] is False
        lg_run.fused_adam(True)                                                                    # idempotent
    finally:
        lg_run.fused_adam(False)
    assert torch.optim.Adam.__init__ is orig
