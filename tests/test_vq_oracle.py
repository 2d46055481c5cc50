"""The numpy oracle of the VecTree nearest-code search against golden vectors produced by the REFERENCE's own
vectree/vq.py (tests/golden/make_golden_vq.py imports /root/reference/vectree/vq.py here; the .npz travels).  No GPU."""
import os

import numpy as np
import pytest

import common
from oracle import vq_oracle

GOLD = os.path.join(common.ROOT, "tests", "golden", "reference_vq.npz")


@pytest.mark.parametrize("name", ["deg2", "deg3", "tiny"])
def test_oracle_reproduces_the_reference_indices(name):
    z = np.load(GOLD)
    ind, gap = vq_oracle.nearest_code(z[f"{name}_x"], z[f"{name}_embed"])
    ref = z[f"{name}_ind"]
    bad = np.nonzero(ind != ref)[0]
    # the reference measures distances in float32 through a matmul; only numerical ties may differ
    assert all(z[f"{name}_gap"][i] < 1e-5 for i in bad), (name, bad[:10])
    assert len(bad) <= 2
    assert np.allclose(gap, z[f"{name}_gap"], atol=1e-9)
    # the quantised rows the reference returned are the codebook rows at those indices
    assert np.array_equal(z[f"{name}_quant"], z[f"{name}_embed"][ref])


def test_golden_can_be_regenerated_when_the_reference_is_present():
    if not os.path.exists("/root/reference/vectree/vq.py"):
        pytest.skip("reference tree absent (GPU box)")
    import subprocess
    import sys
    import tempfile
    src = open(os.path.join(common.ROOT, "tests", "golden", "make_golden_vq.py")).read()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "make_golden_vq.py")
        open(path, "w").write(src)
        subprocess.check_call([sys.executable, path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        new = np.load(os.path.join(td, "reference_vq.npz"))
        old = np.load(GOLD)
        for k in old.files:
            assert np.array_equal(old[k], new[k]), k
