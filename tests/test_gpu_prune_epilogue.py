"""-m gpu: the device-resident prune epilogue (lg_prune_epilogue: radix selects) against the reference's own
formulation -- prune.py:112-128 and scene/gaussian_model.py:776-782 restated in oracle/oracle.py (pinned by
tests/golden/reference_python.npz) -- and against the torch formulation the reference runs."""
import numpy as np
import pytest
import torch

from lightgaussian_amd import prune
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _G:
    def __init__(self, scaling):
        self._s = scaling

    @property
    def get_scaling(self):
        return self._s


def _case(N, seed, zero_frac=0.0, ties=False):
    rng = np.random.default_rng(seed)
    scaling = np.exp(rng.normal(np.log(0.01), 0.7, (N, 3))).astype(np.float32)
    imp = (rng.random(N) ** 3 * 50).astype(np.float32)
    if zero_frac:
        imp[rng.random(N) < zero_frac] = 0.0          # never-hit Gaussians: a large tie at score 0
    if ties:
        imp = np.round(imp)                             # many equal scores
        scaling[:] = 0.01
    return scaling, imp


@pytest.mark.parametrize("N,seed,zero_frac,ties,percent,v_pow", [
    (1, 0, 0, False, 0.66, 0.1), (2, 1, 0, False, 0.5, 0.1), (10, 2, 0, False, 0.7, 0.1), (257, 3, 0.3, False, 0.66, 0.1),
    (5000, 4, 0.8, False, 0.66, 0.1), (5000, 5, 0, True, 0.3, 0.1), (100003, 6, 0.2, False, 0.0, 0.25),
    (100003, 7, 0.2, False, 1.0, 0.1), (3_000_000, 8, 0.3, False, 0.66, 0.1)])
def test_prune_epilogue_matches_reference_formulation(N, seed, zero_frac, ties, percent, v_pow):
    scaling, imp = _case(N, seed, zero_frac, ties)
    st, it = torch.tensor(scaling, device=DEV), torch.tensor(imp, device=DEV)
    v_list, mask, thr = prune.prune_epilogue(_G(st), it, v_pow, percent)
    v = v_list.cpu().numpy(); m = mask.cpu().numpy(); thr = thr.cpu().numpy()

    # order statistics are exact: the same ELEMENT a sort would pick
    volume = (scaling[:, 0] * scaling[:, 1]) * scaling[:, 2]
    assert thr[0] == np.sort(volume)[::-1][int(N * 0.9)]
    assert thr[1] == np.sort(v)[int(percent * (N - 1))]
    # mask: exactly the reference rule applied to the device's own v_list (ties pruned)
    assert np.array_equal(m, oracle.prune_mask(percent, v))
    # v_list: float arithmetic (division, powf, product) within 1e-6 of the numpy restatement ...
    ref = oracle.calculate_v_imp_score(scaling, imp, v_pow)
    assert np.abs(v - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1e-30)
    # ... and the torch formulation the reference runs (same device, same ops) gives the same mask
    tv = prune.calculate_v_imp_score(_G(st), it, v_pow)
    tm = prune.prune_mask(percent, tv)
    assert np.abs(v - tv.cpu().numpy()).max() <= 2e-6 * max(np.abs(ref).max(), 1e-30)
    disagree = int((tm.cpu().numpy().reshape(-1) != m).sum())
    # a last-ulp difference between powf here and torch.pow can move at most the elements at the threshold
    assert disagree <= max(2, int(1e-6 * N)), disagree


def test_prune_epilogue_errors():
    st = torch.rand(4, 3, device=DEV)
    with pytest.raises(Exception):
        prune.prune_epilogue(_G(st[:0]), torch.rand(0, device=DEV), 0.1, 0.5)
    with pytest.raises(ValueError):
        prune.prune_epilogue(_G(st), torch.rand(5, device=DEV), 0.1, 0.5)
    with pytest.raises(Exception):
        prune.prune_epilogue(_G(st), torch.rand(4, device=DEV), 0.1, 1.5)
    with pytest.raises(RuntimeError):
        prune.prune_epilogue(_G(st.cpu()), torch.rand(4), 0.1, 0.5)
