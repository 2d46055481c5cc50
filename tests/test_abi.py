"""The C-ABI library: loads without a GPU, exports every symbol include/lightgaussian.h declares,
and the product path fails loudly (no fallback) when it cannot run.  No compute calls here."""
import ctypes as C
import os
import re

import pytest
import torch

import common
from lightgaussian_amd import _lib

HDR = os.path.join(common.ROOT, "include", "lightgaussian.h")
HDR_DEBUG = os.path.join(common.ROOT, "include", "lightgaussian_debug.h")     # diagnostics for tests/ and tools/: not the drop-in ABI


def _declared_functions(hdr=None):
    if hdr is None:
        return sorted(set(_declared_functions(HDR)) | set(_declared_functions(HDR_DEBUG)))
    src = open(hdr).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|size_t|void|const char\*)\s+\*?(lg_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = _declared_functions()
    for must in ("lg_forward", "lg_forward_count", "lg_backward", "lg_geom_bytes", "lg_img_bytes", "lg_binning_bytes",
                 "lg_backward_scratch_bytes", "lg_score_from_count", "lg_last_error", "lg_abi_version"):
        assert must in names
    assert set(names) == set(_lib.EXPORTS), (sorted(set(names) ^ set(_lib.EXPORTS)))
    # the drop-in header holds no diagnostics, the debug header nothing else (r3 verdict: eight lg_debug_* in the public ABI)
    assert not [n for n in _declared_functions(HDR) if n.startswith("lg_debug_")]
    assert all(n.startswith("lg_debug_") for n in _declared_functions(HDR_DEBUG)) and len(_declared_functions(HDR_DEBUG)) == 8


def test_library_built_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first (hipcc --offload-arch=gfx950)"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/lightgaussian.h but not exported"
    lib.lg_abi_version.restype = C.c_int
    assert lib.lg_abi_version() == 7
    lib.lg_img_bytes.restype = C.c_size_t
    lib.lg_img_bytes.argtypes = [C.c_int32, C.c_int32]
    assert lib.lg_img_bytes(1920, 1080) >= 1920 * 1080 * 8
    lib.lg_backward_scratch_bytes.restype = C.c_size_t
    lib.lg_backward_scratch_bytes.argtypes = [C.c_int32, C.c_int64]
    assert lib.lg_backward_scratch_bytes(1000, 5000) >= 5000 * 9 * 4


def test_struct_layout_matches_header():
    # field order of lg_view / lg_gaussians as declared in the header (ctypes mirrors must follow it)
    src = open(HDR).read()
    view = re.search(r"typedef struct lg_view \{(.*?)\} lg_view;", src, flags=re.S).group(1)
    fields = re.findall(r"\b([a-z_0-9]+);", re.sub(r"/\*.*?\*/", "", view, flags=re.S))
    assert fields == [f[0] for f in _lib.lg_view._fields_]
    gs = re.search(r"typedef struct lg_gaussians \{(.*?)\} lg_gaussians;", src, flags=re.S).group(1)
    fields = re.findall(r"\b([A-Za-z_0-9]+);", re.sub(r"/\*.*?\*/", "", gs, flags=re.S))
    assert fields == [f[0] for f in _lib.lg_gaussians._fields_]


def test_invalid_arguments_rejected_before_any_launch():
    lib = _lib.load()
    assert lib.lg_forward(None, None, None, None, _lib.ALLOC_FN(lambda u, n: 0), None, None, None, None, None, None) == _lib.LG_ERR_INVALID_ARGUMENT
    assert b"null" in lib.lg_last_error()


def test_no_cpu_fallback():
    """CPU tensors must raise, never silently run somewhere else."""
    from lightgaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False, False)
    r = GaussianRasterizer(rs)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1), colors_precomp=torch.ones(4, 3),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="excatly one"):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))


def test_product_never_imports_the_oracle():
    import glob
    for path in glob.glob(os.path.join(common.ROOT, "lightgaussian_amd", "**", "*.py"), recursive=True) + \
            glob.glob(os.path.join(common.ROOT, "diff_gaussian_rasterization", "*.py")):
        txt = open(path).read()
        assert "oracle" not in txt.replace("(the reference has no CPU", ""), path
    for path in glob.glob(os.path.join(common.ROOT, "lightgaussian_amd", "csrc", "*")):
        assert "lg_oracle" not in open(path).read(), path


def test_the_library_has_no_process_wide_setters_and_the_segment_length_travels_with_the_view():
    """ABI v5 (r2 verdict): lg_set_segment_length / lg_set_long_tile_mode are gone -- the segment length is a field of lg_view,
    the long-tile mode two flag bits -- and the buffer size is a pure function of its arguments (no device needed here)."""
    lib = _lib.load()
    assert not hasattr(lib, "lg_set_segment_length") and not hasattr(lib, "lg_set_long_tile_mode")
    for name in ("lg_set_segment_length", "lg_set_long_tile_mode"):
        with pytest.raises(AttributeError):
            getattr(C.CDLL(_lib.LIB_PATH), name)
    assert [f[0] for f in _lib.lg_view._fields_][-2:] == ["segment_length", "count_sum"]
    for n in (0, 1, 63, 64, 100000):                                     # visibility bytes live inside the geom buffer (host-only query)
        off = lib.lg_geom_visible_offset(n)
        assert off % 16 == 0 and off + n <= lib.lg_geom_bytes(n)
    a, b = lib.lg_binning_bytes(100000, 640, 480, 0), lib.lg_binning_bytes(100000, 640, 480, 512)
    assert a == b > 0                                                   # 0 = the default of 512
    assert lib.lg_binning_bytes(100000, 640, 480, 64) > a               # more checkpoint records for shorter segments
    assert lib.lg_binning_bytes(100000, 640, 480, 100) == 0             # not a multiple of 64
    assert lib.lg_binning_bytes(100000, 640, 480, 32) == 0              # below 64
    assert lib.lg_binning_bytes(100000, 640, 480, 0) == a               # ... and asking twice changes nothing
    # a view with a bad segment length or contradictory long-tile flags is refused before any launch
    v = _lib.lg_view(8, 8, 1.0, 1.0, None, 1.0, None, None, 0, None, 0, 0, 100)
    g = _lib.lg_gaussians(0, 0, None, None, None, None, None, None, None, None)
    cb = _lib.ALLOC_FN(lambda u, n: 0)
    assert lib.lg_forward(C.byref(v), C.byref(g), None, None, cb, None, None, None, None, None, None) == _lib.LG_ERR_INVALID_ARGUMENT
    assert b"segment_length" in lib.lg_last_error()
    v = _lib.lg_view(8, 8, 1.0, 1.0, None, 1.0, None, None, 0, None, 0, _lib.FLAG_LONG_SERIAL | _lib.FLAG_LONG_PARALLEL, 0)
    assert lib.lg_forward(C.byref(v), C.byref(g), None, None, cb, None, None, None, None, None, None) == _lib.LG_ERR_INVALID_ARGUMENT
    assert b"exclude each other" in lib.lg_last_error()


def test_options_are_resolved_per_call_and_per_thread_never_by_mutating_the_defaults():
    """rasterizer.options(...) is thread-local and nests; an `options=` argument wins; the process defaults stay untouched
    (r2 verdict: prune_list_sharded / _ViewRunner used to flip module-level switches under other threads' feet)."""
    import threading
    from lightgaussian_amd import rasterizer as R
    base = dict(R._OPTIONS)
    seen = {}

    def other():
        seen["other"] = R.resolve_options()

    with R.options(sync_free=True, skip_color_in_count=True):
        with R.options(tag=7, long_tiles="serial"):
            inner = R.resolve_options({"segment_length": 128})
            t = threading.Thread(target=other); t.start(); t.join()
        mid = R.resolve_options()
    assert inner["sync_free"] is True and inner["skip_color_in_count"] is True and inner["tag"] == 7
    assert inner["long_tiles"] == "serial" and inner["segment_length"] == 128
    assert mid["long_tiles"] == base["long_tiles"] and "tag" not in mid and mid["sync_free"] is True
    assert seen["other"] == base                                          # the other thread saw the defaults
    assert R.resolve_options() == base and R._OPTIONS == base
    with pytest.raises(KeyError):
        R.options(no_such_knob=1)
    with pytest.raises(ValueError):
        R.resolve_options({"segment_length": 100})
    with pytest.raises(ValueError):
        R.set_option("long_tiles", "sometimes")
    assert R.set_option("long_tiles", "parallel") == base["long_tiles"]   # returns the previous default
    assert R.set_option("long_tiles", base["long_tiles"]) == "parallel"
    # a PendingBatch keeps the status words of ITS forwards only, tagged
    b = R.PendingBatch()
    b.add(None, ("k",), 3); b.add(None, ("k",), 5)
    assert b.resolve() == [(3, False), (5, False)] and b.resolve() == []


def test_prune_pass_and_view_runner_do_not_touch_the_process_defaults():
    import inspect
    from lightgaussian_amd import prune, parallel, dp
    for mod in (prune, parallel, dp):
        src = inspect.getsource(mod)
        assert "set_option(" not in src and "_OPTIONS[" not in src, mod.__name__
