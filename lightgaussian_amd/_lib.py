"""ctypes binding of the C ABI declared in include/lightgaussian.h.

PyTorch is used only for device memory and streams; every compute call goes through
liblightgaussian_hip.so.  There is NO fallback: if the library is missing or a GPU call
fails, this module raises -- it never routes to a CPU/PyTorch implementation.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIGHTGAUSSIAN_HIP_LIB: A/B builds of the same library on one GPU box (tools/gpu_ab.sh); always an in-tree HIP build
LIB_PATH = os.environ.get("LIGHTGAUSSIAN_HIP_LIB") or os.path.join(_HERE, "liblightgaussian_hip.so")

LG_OK = 0
LG_ERR_INVALID_ARGUMENT = -1
LG_ERR_DEVICE = -2
LG_ERR_ALLOC = -3
LG_ERR_PREFILTERED = -4

WEIGHT_ONE, WEIGHT_OPACITY, WEIGHT_ALPHA, WEIGHT_ALPHA_T = 0, 1, 2, 3
FLAG_DEBUG, FLAG_FAST_EXP, FLAG_PROFILE, FLAG_RAW_PARAMS, FLAG_SKIP_COLOR, FLAG_L1_ONLY = 1, 2, 4, 8, 16, 32
FLAG_NARROW_KEY, FLAG_SORT_ALL_BITS, FLAG_K1_LDS = 64, 128, 256
FLAG_LONG_SERIAL, FLAG_LONG_PARALLEL = 512, 1024
FLAG_SAVE_SH_JACOBIAN = 2048
FLAG_COUNT_WIDE_BAND = 8192
ABI_VERSION = 7     # include/lightgaussian.h LG_ABI_VERSION this binding was written against (load() refuses another)

EXPORTS = ["lg_geom_bytes", "lg_img_bytes", "lg_binning_bytes", "lg_backward_scratch_bytes", "lg_forward",
           "lg_forward_count", "lg_backward", "lg_score_from_count", "lg_abi_version", "lg_last_error",
           "lg_profile_read", "lg_profile_reset", "lg_last_stats", "lg_debug_reduce9", "lg_loss_state_bytes",
           "lg_loss_forward", "lg_loss_backward", "lg_prune_scratch_bytes", "lg_prune_epilogue", "lg_knn_scratch_bytes",
           "lg_knn3_mean_dist2", "lg_ordered_sum", "lg_forward_bounded", "lg_select_mask", "lg_compact_scratch_bytes",
           "lg_compact_plan", "lg_compact_rows", "lg_vq_scratch_bytes", "lg_vq_nearest", "lg_debug_sort_temp_bytes",
           "lg_debug_sort_keys", "lg_build_id", "lg_backward_chunked", "lg_debug_activations", "lg_view_status",
           "lg_debug_sort_orphan", "lg_debug_last_contributor", "lg_debug_tile_lists", "lg_geom_visible_offset",
           "lg_sh_grad_from_rgb", "lg_debug_view_meta"]


class lg_view(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("bg", C.c_void_p), ("scale_modifier", C.c_float), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("sh_degree", C.c_int32), ("campos", C.c_void_p),
                ("prefiltered", C.c_int32), ("flags", C.c_uint32), ("segment_length", C.c_int32), ("count_sum", C.c_void_p)]


class lg_gaussians(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("means3D", C.c_void_p), ("shs", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("shs_rest", C.c_void_p)]


class lg_kernel_time(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_ms", C.c_double), ("launches", C.c_int64)]


class lg_stats(C.Structure):
    _fields_ = [("num_rendered", C.c_int64), ("num_visible", C.c_int64)]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
CHUNK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32)


def build(force=False, verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", src_dir] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL, stderr=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load():
    """Load the native library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X rasterizer has no CPU/PyTorch fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    vp, P = C.c_void_p, C.POINTER
    lib.lg_abi_version.restype = C.c_int; lib.lg_abi_version.argtypes = []
    if lib.lg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI {lib.lg_abi_version()}, this binding needs {ABI_VERSION}: rebuild the library "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
    lib.lg_geom_bytes.restype = C.c_size_t; lib.lg_geom_bytes.argtypes = [C.c_int32]
    lib.lg_img_bytes.restype = C.c_size_t; lib.lg_img_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.lg_binning_bytes.restype = C.c_size_t; lib.lg_binning_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    lib.lg_backward_scratch_bytes.restype = C.c_size_t; lib.lg_backward_scratch_bytes.argtypes = [C.c_int32, C.c_int64]
    lib.lg_forward.restype = C.c_int
    lib.lg_forward.argtypes = [P(lg_view), P(lg_gaussians), vp, vp, ALLOC_FN, vp, vp, vp, P(vp), P(C.c_int64), vp]
    lib.lg_forward_count.restype = C.c_int
    lib.lg_forward_count.argtypes = [P(lg_view), P(lg_gaussians), vp, vp, ALLOC_FN, vp, C.c_int32, vp, vp, vp, vp, P(vp),
                                     P(C.c_int64), vp]
    lib.lg_backward.restype = C.c_int
    lib.lg_backward.argtypes = [P(lg_view), P(lg_gaussians), vp, vp, vp, vp, C.c_int64, vp] + [vp] * 9 + [vp, vp]
    lib.lg_backward_chunked.restype = C.c_int
    lib.lg_backward_chunked.argtypes = [P(lg_view), P(lg_gaussians), vp, vp, vp, vp, C.c_int64, vp] + [vp] * 9 + [vp, vp, C.c_int32, CHUNK_FN, vp]
    lib.lg_score_from_count.restype = C.c_int
    lib.lg_score_from_count.argtypes = [C.c_int32, vp, vp, vp, vp]
    lib.lg_loss_state_bytes.restype = C.c_size_t; lib.lg_loss_state_bytes.argtypes = [C.c_int32] * 3
    lib.lg_loss_forward.restype = C.c_int
    lib.lg_loss_forward.argtypes = [C.c_int32] * 3 + [vp, vp, vp, vp, C.c_uint32, vp]
    lib.lg_loss_backward.restype = C.c_int
    lib.lg_loss_backward.argtypes = [C.c_int32] * 3 + [vp, vp, vp, vp, C.c_float, vp, C.c_float, vp, C.c_uint32, vp]
    lib.lg_prune_scratch_bytes.restype = C.c_size_t; lib.lg_prune_scratch_bytes.argtypes = [C.c_int32]
    lib.lg_prune_epilogue.restype = C.c_int
    lib.lg_prune_epilogue.argtypes = [C.c_int32, vp, vp, C.c_float, C.c_double, vp, vp, vp, vp, C.c_uint32, vp]
    lib.lg_knn_scratch_bytes.restype = C.c_size_t; lib.lg_knn_scratch_bytes.argtypes = [C.c_int32]
    lib.lg_knn3_mean_dist2.restype = C.c_int
    lib.lg_knn3_mean_dist2.argtypes = [C.c_int32, vp, vp, vp, C.c_uint32, vp]
    lib.lg_sh_grad_from_rgb.restype = C.c_int
    lib.lg_sh_grad_from_rgb.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_int64, C.c_float, C.c_int32, vp, vp, vp]
    lib.lg_ordered_sum.restype = C.c_int
    lib.lg_ordered_sum.argtypes = [C.c_int32, C.c_int64, vp, C.c_int64, vp, vp]
    lib.lg_forward_bounded.restype = C.c_int
    lib.lg_forward_bounded.argtypes = [P(lg_view), P(lg_gaussians), vp, vp, vp, C.c_int64, C.c_float, C.c_int32, vp, vp, vp, vp, vp,
                                       P(C.c_uint32 * 4), vp]
    lib.lg_select_mask.restype = C.c_int
    lib.lg_select_mask.argtypes = [C.c_int32, vp, C.c_int64, vp, vp, vp, vp]
    lib.lg_compact_scratch_bytes.restype = C.c_size_t; lib.lg_compact_scratch_bytes.argtypes = [C.c_int32]
    lib.lg_compact_plan.restype = C.c_int
    lib.lg_compact_plan.argtypes = [C.c_int32, vp, vp, vp, vp, vp]
    lib.lg_compact_rows.restype = C.c_int
    lib.lg_compact_rows.argtypes = [C.c_int32, vp, C.c_int32, P(vp), P(vp), P(C.c_int32), vp]
    lib.lg_vq_scratch_bytes.restype = C.c_size_t; lib.lg_vq_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.lg_vq_nearest.restype = C.c_int
    lib.lg_vq_nearest.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_uint32, vp]
    lib.lg_debug_sort_temp_bytes.restype = C.c_size_t; lib.lg_debug_sort_temp_bytes.argtypes = [C.c_int64]
    lib.lg_debug_sort_keys.restype = C.c_int
    lib.lg_debug_sort_keys.argtypes = [C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, vp]
    lib.lg_debug_activations.restype = C.c_int; lib.lg_debug_activations.argtypes = [C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.lg_debug_reduce9.restype = C.c_int; lib.lg_debug_reduce9.argtypes = [vp, vp, vp]
    lib.lg_abi_version.restype = C.c_int; lib.lg_abi_version.argtypes = []
    lib.lg_last_error.restype = C.c_char_p; lib.lg_last_error.argtypes = []
    lib.lg_build_id.restype = C.c_char_p; lib.lg_build_id.argtypes = []
    lib.lg_view_status.restype = C.c_int; lib.lg_view_status.argtypes = [vp, C.c_int32, P(C.c_uint32 * 4), vp]
    lib.lg_debug_last_contributor.restype = C.c_int
    lib.lg_debug_last_contributor.argtypes = [P(lg_view), C.c_int32, vp, vp, vp, C.c_int64, vp, vp]
    lib.lg_debug_sort_orphan.restype = C.c_int; lib.lg_debug_sort_orphan.argtypes = [C.c_int64, vp, vp, vp, vp]
    lib.lg_geom_visible_offset.restype = C.c_size_t; lib.lg_geom_visible_offset.argtypes = [C.c_int32]
    lib.lg_debug_view_meta.restype = C.c_int; lib.lg_debug_view_meta.argtypes = [P(lg_view), vp, C.c_int64, vp, vp]
    lib.lg_debug_tile_lists.restype = C.c_int; lib.lg_debug_tile_lists.argtypes = [P(lg_view), vp, C.c_int64, vp, vp, vp]
    lib.lg_profile_read.restype = C.c_int; lib.lg_profile_read.argtypes = [P(lg_kernel_time), C.c_int]
    lib.lg_profile_reset.restype = None; lib.lg_profile_reset.argtypes = []
    lib.lg_last_stats.restype = C.c_int; lib.lg_last_stats.argtypes = [P(lg_stats)]
    _lib = lib
    return lib


def check(rc):
    """Map C error codes onto the reference's Python-side behaviour: invalid argument combos ->
    Exception, device errors -> RuntimeError."""
    if rc == LG_OK:
        return
    msg = load().lg_last_error().decode("utf-8", "replace")
    if rc == LG_ERR_INVALID_ARGUMENT:
        raise Exception(msg)
    raise RuntimeError(f"lightgaussian_hip error {rc}: {msg}")


def build_id():
    """Identity of the kernel sources the loaded library was built from (see include/lightgaussian.h lg_build_id)."""
    return load().lg_build_id().decode()


def profile_read():
    """Per-kernel hipEvent totals recorded under FLAG_PROFILE: {name: (total_ms, launches)}."""
    lib = load()
    arr = (lg_kernel_time * 32)()
    n = lib.lg_profile_read(arr, 32)
    return {arr[i].name.decode(): (arr[i].total_ms, arr[i].launches) for i in range(min(n, 32))}


def profile_reset():
    load().lg_profile_reset()


def last_stats():
    s = lg_stats()
    load().lg_last_stats(C.byref(s))
    return {"num_rendered": s.num_rendered, "num_visible": s.num_visible}
