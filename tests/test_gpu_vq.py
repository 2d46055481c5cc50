"""-m gpu: the MFMA nearest-code search (lg_vq_nearest) against the golden vectors of the reference's own vectree/vq.py and
against the numpy oracle at the reference's full shapes (8192-entry codebook, 27 / 48 dimensions, chunks of 8192 rows as
vectree/vectree.py:87-101 feeds them).  Index-exact; a mismatch is tolerated only where best and second-best code are a
numerical tie (gap < 1e-6 of the distance scale)."""
import os

import numpy as np
import pytest
import torch

import common
from lightgaussian_amd import vq
from oracle import vq_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(common.ROOT, "tests", "golden", "reference_vq.npz")


@pytest.mark.parametrize("name", ["deg2", "deg3", "tiny"])
def test_golden_vectors_of_the_reference_vq(name):
    z = np.load(GOLD)
    x, e = torch.tensor(z[f"{name}_x"], device=DEV), torch.tensor(z[f"{name}_embed"], device=DEV)
    ind = vq.nearest_code(x, e).cpu().numpy()
    bad = np.nonzero(ind != z[f"{name}_ind"])[0]
    assert all(z[f"{name}_gap"][i] < 1e-5 for i in bad), (name, bad[:10], len(bad))
    assert len(bad) <= 2
    q, ind2 = vq.quantize(x.unsqueeze(0), e.unsqueeze(0))          # the [h, n, d] form the reference calls with
    assert ind2.shape == (1, x.shape[0]) and ind2.dtype == torch.int64
    if len(bad) == 0:
        assert np.array_equal(q[0].cpu().numpy(), z[f"{name}_quant"])


@pytest.mark.parametrize("n,d,K,seed", [(8192, 27, 8192, 0), (8192, 48, 8192, 1), (20000, 27, 8192, 2), (1000, 12, 100, 3),
                                        (129, 3, 7, 4), (5000, 63, 1000, 5), (1, 27, 8192, 6), (4097, 1, 3, 7)])
def test_full_size_against_the_oracle(n, d, K, seed):
    rng = np.random.default_rng(seed)
    e = (rng.standard_normal((K, d)) * 0.3).astype(np.float32)
    x = (e[rng.integers(0, K, n)] + 0.1 * rng.standard_normal((n, d))).astype(np.float32)
    ref, gap = vq_oracle.nearest_code(x, e)
    ind = vq.nearest_code(torch.tensor(x, device=DEV), torch.tensor(e, device=DEV)).cpu().numpy()
    bad = np.nonzero(ind != ref)[0]
    scale = float(np.abs(x).max() * np.sqrt(d)) + 1e-30
    assert all(gap[i] < 1e-5 * scale for i in bad), (bad[:10], [gap[i] for i in bad[:10]])
    assert len(bad) <= max(2, n // 5000)
    assert ind.min() >= 0 and ind.max() < K


def test_ties_go_to_the_lowest_index_and_errors():
    e = torch.zeros(300, 27, device=DEV)
    e[7] = 1.0; e[130] = 1.0; e[299] = 1.0                      # three identical codes in different 128-code chunks / tiles
    x = torch.ones(100, 27, device=DEV)
    assert torch.equal(vq.nearest_code(x, e), torch.full((100,), 7, device=DEV))
    assert torch.equal(vq.nearest_code(torch.zeros(5, 27, device=DEV), e), torch.zeros(5, dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError):
        vq.nearest_code(torch.zeros(5, 27), torch.zeros(3, 27))
    with pytest.raises(ValueError):
        vq.nearest_code(x, torch.zeros(3, 26, device=DEV))
    with pytest.raises(Exception):
        vq.nearest_code(torch.zeros(5, 64, device=DEV), torch.zeros(3, 64, device=DEV))
    assert vq.nearest_code(torch.zeros(0, 27, device=DEV), e).shape == (0,)
