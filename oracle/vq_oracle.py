"""CPU oracle of the VecTree nearest-code search (TEST INFRASTRUCTURE -- only tests/ may import this).

Restates vectree/vq.py:262-266 of the reference (EuclideanCodebook.forward: dist = -torch.cdist(flatten, embed, p=2);
embed_ind = dist.argmax(-1), gumbel_sample at temperature 0) in numpy float64: argmin of the squared Euclidean distance,
first minimum on ties.  Pinned: tests/test_vq_oracle.py checks it against tests/golden/reference_vq.npz, which
tests/golden/make_golden_vq.py produced by running the reference's own EuclideanCodebook."""
import numpy as np


def nearest_code(x, embed, block=4096):
    x = np.asarray(x, np.float64)
    e = np.asarray(embed, np.float64)
    en = (e * e).sum(1)
    out = np.empty(x.shape[0], np.int64)
    gap = np.empty(x.shape[0], np.float64)
    for lo in range(0, x.shape[0], block):
        xs = x[lo:lo + block]
        d2 = (xs * xs).sum(1)[:, None] - 2.0 * xs @ e.T + en[None, :]
        out[lo:lo + block] = d2.argmin(1)
        if e.shape[0] > 1:
            part = np.partition(d2, 1, axis=1)[:, :2]
            gap[lo:lo + block] = np.sqrt(np.maximum(part[:, 1], 0)) - np.sqrt(np.maximum(part[:, 0], 0))
        else:
            gap[lo:lo + block] = np.inf
    return out, gap
