"""CPU restatement of the reference's simple-knn result -- TEST INFRASTRUCTURE ONLY (tests/ and smoke() only).

submodules/simple-knn/simple_knn.cu:147-183 (boxMeanDist): for every point the three smallest squared distances to the
other points (self excluded BY INDEX, so coincident points count with distance 0), kept in an ascending triple that
starts at FLT_MAX, and dists[i] = (best[0] + best[1] + best[2]) / 3.0f.  The Morton boxes only prune the search; the
result is the exact 3-nearest-neighbour set, which is what this file computes (scipy cKDTree, float64 distances from
float32 coordinates, float32 sum like the kernel).

PARITY: the reference vendors the CUDA source of this component, so its semantics are pinned by the source itself; it
cannot be compiled here (no nvcc), hence no golden vectors from the reference binary -- the known-answer cases of
tests/test_knn.py are derived by hand from the lines cited above.
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def dist_cuda2(points):
    p32 = np.ascontiguousarray(points, dtype=np.float32)
    P = p32.shape[0]
    out = np.zeros(P, dtype=np.float32)
    if P == 0:
        return out
    best = np.full((P, 3), FLT_MAX, dtype=np.float32)
    if P > 1:
        from scipy.spatial import cKDTree
        p = p32.astype(np.float64)
        k = min(4, P)
        _, idx = cKDTree(p).query(p, k=k)
        idx = idx.reshape(P, k)
        for i in range(P):
            # drop SELF by index (a coincident twin may be returned before the point itself)
            nb = [j for j in idx[i] if j != i][:3]
            if len(nb) < min(3, P - 1):                      # self was not among the k hits (many coincident points)
                d_all = ((p - p[i]) ** 2).sum(1); d_all[i] = np.inf
                nb = list(np.argsort(d_all, kind="stable")[:min(3, P - 1)])
            d = p32[nb] - p32[i]
            d2 = np.sort((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32))
            best[i, :len(d2)] = d2
    with np.errstate(over="ignore"):
        out = ((best[:, 0] + best[:, 1]) + best[:, 2]) / np.float32(3.0)
    return out.astype(np.float32)
