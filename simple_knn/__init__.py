"""Import shim for the reference's `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20).

simple-knn is NOT on the north-star hot path (init-only, used once by
GaussianModel.create_from_pcd, scene/gaussian_model.py:152-156).  It is provided so that the reference's
trainers import on a ROCm box with this repo on PYTHONPATH; see simple_knn/_C.py."""
