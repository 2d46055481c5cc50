"""distCUDA2 for ROCm: drop-in for the reference's simple-knn extension (submodules/simple-knn/spatial.cu:15-27 ->
simple_knn.cu:185-221), consumed at scene/gaussian_model.py:152-156:

    dist2 = torch.clamp_min(distCUDA2(torch.from_numpy(np.asarray(pcd.points)).float().cuda()), 0.0000001)

mean of the squared distances to the 3 nearest other points, exact.  Runs on the HIP library
(lg_knn3_mean_dist2: multi-level uniform grid, one radix sort per level; one 4-byte read-back at the end says whether a sort gave up); there is no CPU / PyTorch
fallback -- CPU tensors raise, like the CUDA extension does.
"""
import ctypes as C

import torch


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float32 on the GPU -> [P] float32."""
    from lightgaussian_amd import _lib
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("distCUDA2 expects a [P,3] tensor")
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 runs on the MI355X HIP library only (no CPU fallback); pass a .cuda() tensor")
    pts = points.detach().float().contiguous()
    P = pts.shape[0]
    means = torch.full((P,), 0.0, dtype=torch.float32, device=pts.device)    # spatial.cu:21
    if P == 0:
        return means
    lib = _lib.load()
    scratch = torch.empty(lib.lg_knn_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.lg_knn3_mean_dist2(P, pts.data_ptr(), means.data_ptr(), scratch.data_ptr(), 0,
                                      C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)))
    return means
