"""distCUDA2 for ROCm, written against the reference's call site only
(scene/gaussian_model.py:152-156:  dist2 = torch.clamp_min(distCUDA2(points.float().cuda()), 0.0000001)).

Semantics of the reference extension (submodules/simple-knn/simple_knn.cu:147-183,185-221): for every point the
MEAN of the squared Euclidean distances to its 3 nearest other points.  The CUDA original finds them with a
Morton-order sweep; this is an exact, chunked brute force on the tensor's own device (torch ops, runs on HIP):
O(N^2) distance evaluations in tiles of `chunk` x N, which for SfM-sized clouds (1e4..1e6 points, called once at
start-up) is seconds at most on an MI355X.  Init-only plumbing -- deliberately not a hand-written kernel.
"""
import torch


def distCUDA2(points: torch.Tensor, chunk: int = 4096) -> torch.Tensor:
    """points [P,3] float -> [P] float: mean squared distance to the 3 nearest neighbours (self excluded)."""
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("distCUDA2 expects a [P,3] tensor")
    pts = points.float().contiguous()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    k = min(4, P)  # self + 3 neighbours
    sq = (pts * pts).sum(1)
    for s in range(0, P, chunk):
        q = pts[s:s + chunk]
        d2 = (sq[s:s + chunk, None] + sq[None, :] - 2.0 * (q @ pts.t())).clamp_min_(0.0)
        idx = torch.arange(s, min(s + chunk, P), device=pts.device)
        d2[torch.arange(idx.numel(), device=pts.device), idx] = float("inf")  # exclude self exactly
        nn = torch.topk(d2, k - 1, dim=1, largest=False).values if k > 1 else torch.zeros(idx.numel(), 1, device=pts.device)
        # exact recomputation of the selected distances (the expanded form above loses precision for near-duplicates)
        sel = torch.topk(d2, k - 1, dim=1, largest=False).indices if k > 1 else None
        if sel is not None:
            diff = q[:, None, :] - pts[sel]
            nn = (diff * diff).sum(-1)
        # the reference divides the sum of the 3 best by 3 (simple_knn.cu:182) even if fewer than 3 neighbours exist
        out[s:s + chunk] = nn.sum(1) / 3.0
    return out
