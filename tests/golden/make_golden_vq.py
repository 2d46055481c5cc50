"""Golden vectors for the VecTree nearest-code search, produced by the REFERENCE's own code: imports
/root/reference/vectree/vq.py (EuclideanCodebook, unmodified) in this container and runs its forward in eval mode --
dist = -torch.cdist(flatten, embed, p=2); embed_ind = dist.argmax(-1) (vectree/vq.py:262-266) -- on seeded inputs with the
two feature widths of vectree/vectree.py:31-35 (27 = SH degree 2, 48 = degree 3).  Stores inputs, the reference's indices and
the gap between the best and the second-best distance (float64) so that the parity test can tell a genuine mismatch from a
numerical tie.  Run:  python tests/golden/make_golden_vq.py   (needs /root/reference; the committed .npz travels)."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/vectree"
sys.path.insert(0, REF)
from vq import EuclideanCodebook  # noqa: E402

out = {}
gen = torch.Generator().manual_seed(20250924)
for name, n, d, K in (("deg2", 1500, 27, 512), ("deg3", 700, 48, 384), ("tiny", 65, 3, 5)):
    embed = torch.randn(K, d, generator=gen) * 0.3
    # feature rows: noisy copies of random codes (the regime after k-means) plus some far outliers
    x = embed[torch.randint(0, K, (n,), generator=gen)] + 0.12 * torch.randn(n, d, generator=gen)
    x[::17] = torch.randn(x[::17].shape, generator=gen) * 1.5
    cb = EuclideanCodebook(dim=d, codebook_size=K, kmeans_init=False, threshold_ema_dead_code=0)
    cb.embed.data.copy_(embed.unsqueeze(0))
    cb.eval()
    with torch.no_grad():
        quant, ind = cb(x.unsqueeze(0))
    ind = ind.reshape(-1)
    d64 = torch.cdist(x.double(), embed.double())
    two = torch.topk(d64, 2, dim=1, largest=False).values
    out[f"{name}_x"] = x.numpy(); out[f"{name}_embed"] = embed.numpy()
    out[f"{name}_ind"] = ind.numpy().astype(np.int64)
    out[f"{name}_gap"] = (two[:, 1] - two[:, 0]).numpy()
    out[f"{name}_quant"] = quant.reshape(n, d).numpy()
    assert torch.equal(ind, d64.argmin(1)) or (two[:, 1] - two[:, 0]).min() < 1e-6
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vq.npz"), **out)
print({k: v.shape for k, v in out.items()})
