"""Top-level drop-in module: `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` (gaussian_renderer/__init__.py:14-17 of the reference) resolves here when the
repo root is on sys.path.  Implementation: lightgaussian_amd/rasterizer.py -> liblightgaussian_hip.so."""
from lightgaussian_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)
