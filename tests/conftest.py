import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _deterministic_long_tile_mode(request):
    """GPU tests: long tiles of training renders take the SERIAL walk unless a test asks otherwise.  The library default ("auto")
    switches to the parallel kernels once any view of the process has had a list longer than one segment -- correct either way,
    but two renders of one scene taken before and after that switch differ in the last float bits (regrouped transmittance
    products), and several tests compare renders bit for bit.  tests/test_gpu_long_tiles.py exercises "parallel" and "auto"."""
    if "gpu" not in request.keywords:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from lightgaussian_amd import rasterizer
    rasterizer.set_option("long_tiles", "serial")
    yield
    rasterizer.set_option("long_tiles", "serial")
