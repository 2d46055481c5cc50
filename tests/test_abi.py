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


def _declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|size_t|void|const char\*)\s+\*?(lg_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = _declared_functions()
    for must in ("lg_forward", "lg_forward_count", "lg_backward", "lg_geom_bytes", "lg_img_bytes", "lg_binning_bytes",
                 "lg_backward_scratch_bytes", "lg_score_from_count", "lg_last_error", "lg_abi_version"):
        assert must in names
    assert set(names) == set(_lib.EXPORTS), (sorted(set(names) ^ set(_lib.EXPORTS)))


def test_library_built_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first (hipcc --offload-arch=gfx950)"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/lightgaussian.h but not exported"
    lib.lg_abi_version.restype = C.c_int
    assert lib.lg_abi_version() == 4
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


def test_process_wide_setters_return_the_previous_value_and_reject_nonsense():
    """lg_set_segment_length / lg_set_long_tile_mode (no device needed: host state only)."""
    lib = _lib.load()
    prev = lib.lg_set_segment_length(128)
    try:
        assert lib.lg_set_segment_length(100) == 128          # not a multiple of 64: ignored ...
        assert lib.lg_set_segment_length(32) == 128           # ... as is anything below 64
        assert lib.lg_set_segment_length(256) == 128
        assert lib.lg_binning_bytes(100000, 640, 480) > 0
    finally:
        lib.lg_set_segment_length(prev)
    mode = lib.lg_set_long_tile_mode(0)                       # 0 serial, 1 auto (default), 2 parallel
    try:
        assert mode in (0, 1, 2)
        assert lib.lg_set_long_tile_mode(2) == 0
        assert lib.lg_set_long_tile_mode(7) == 2 and lib.lg_set_long_tile_mode(-1) == 2   # out of range: unchanged
        assert lib.lg_set_long_tile_mode(0) == 2
    finally:
        lib.lg_set_long_tile_mode(0 if mode == 1 else mode)  # (mode 1 touches pinned host memory: not from a test without a device)
