"""Loads the REFERENCE's own modules, unmodified, on top of this repo's shims (test infrastructure).

    /root/reference/gaussian_renderer/__init__.py   render(), count_render()        (imports diff_gaussian_rasterization)
    /root/reference/scene/gaussian_model.py          GaussianModel                   (imports simple_knn._C, plyfile, icecream)
    /root/reference/prune.py                         prune_list, calculate_v_imp_score
with  diff_gaussian_rasterization / simple_knn  resolving to THIS repo (the drop-in shims) and the two pure-Python
dependencies that are not installed here (plyfile, icecream) stubbed.  `scene/__init__.py` (dataset readers, PIL, COLMAP
loaders -- off the path) is not executed: a bare package object with the reference's __path__ stands in, so that
`scene.gaussian_model` is the reference's file itself."""
import importlib
import os
import sys
import types

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def available():
    return os.path.exists(os.path.join(REF, "gaussian_renderer", "__init__.py"))


def load():
    """Returns (gaussian_renderer, gaussian_model, prune) modules of the reference."""
    for name in ("plyfile", "icecream"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.PlyData = m.PlyElement = type("Stub", (), {})
            m.ic = lambda *a, **k: (a[0] if a else None)
            sys.modules[name] = m
    # repo first (shims), reference after it (utils/, arguments/, vectree/, gaussian_renderer/, prune.py)
    for p in (REF, ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    if "scene" not in sys.modules or getattr(sys.modules["scene"], "__file__", "stub") != "stub":
        pkg = types.ModuleType("scene")
        pkg.__path__ = [os.path.join(REF, "scene")]
        pkg.__file__ = "stub"
        sys.modules["scene"] = pkg
    gm = importlib.import_module("scene.gaussian_model")
    sys.modules["scene"].GaussianModel = gm.GaussianModel
    sys.modules["scene"].Scene = type("Scene", (), {})
    import diff_gaussian_rasterization
    import simple_knn._C
    assert os.path.abspath(diff_gaussian_rasterization.__file__).startswith(ROOT), "the shim must win over any installed extension"
    assert os.path.abspath(simple_knn._C.__file__).startswith(ROOT)
    gr = importlib.import_module("gaussian_renderer")
    assert os.path.abspath(gr.__file__).startswith(REF), gr.__file__
    pr = importlib.import_module("prune")
    assert os.path.abspath(pr.__file__).startswith(REF), pr.__file__
    return gr, gm, pr
