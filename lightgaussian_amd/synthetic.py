"""Frozen synthetic scene + camera generator.

Everything here is host-side PyTorch on CPU tensors; callers move the result to the GPU.
Camera matrices follow the reference's conventions exactly:
  world_view_transform = getWorld2View2(R, T).T          scene/cameras.py:70-72, utils/graphics_utils.py:45-55
  projection_matrix    = getProjectionMatrix(...).T      scene/cameras.py:73-79, utils/graphics_utils.py:58-76
  full_proj_transform  = world_view @ projection         scene/cameras.py:80-84
  camera_center        = inverse(world_view)[3, :3]      scene/cameras.py:85
Gaussian parameter tensors have the layouts of GaussianModel's raw parameters
(scene/gaussian_model.py:98-118): _xyz [N,3], _features_dc [N,1,3], _features_rest [N,M-1,3],
_scaling [N,3] (log), _rotation [N,4] (r,x,y,z, un-normalised), _opacity [N,1] (logit).
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

SEED = 20250103
SH_C0 = 0.28209479177387814


def RGB2SH(rgb):
    """utils/sh_utils.py:123-124"""
    return (rgb - 0.5) / SH_C0


@dataclass
class MiniCam:
    """Same attribute surface as scene/cameras.py:88-109 MiniCam (what render() reads)."""
    image_width: int
    image_height: int
    FoVy: float
    FoVx: float
    znear: float
    zfar: float
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    camera_center: torch.Tensor

    def to(self, device):
        return MiniCam(self.image_width, self.image_height, self.FoVy, self.FoVx, self.znear, self.zfar,
                       self.world_view_transform.to(device), self.full_proj_transform.to(device),
                       self.camera_center.to(device))


def world2view(R, t):
    """getWorld2View2 with translate=0, scale=1 (utils/graphics_utils.py:45-55)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY):
    """utils/graphics_utils.py:58-76"""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def make_camera(R, T, width, height, FoVx, FoVy, znear=0.01, zfar=100.0):
    """R = camera-to-world rotation, T = world-to-camera translation (COLMAP convention used by
    scene/dataset_readers.py); returns a MiniCam on CPU."""
    wv = torch.tensor(world2view(R, T)).transpose(0, 1)
    proj = projection_matrix(znear, zfar, FoVx, FoVy).transpose(0, 1)
    full = wv.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    center = wv.inverse()[3, :3]
    return MiniCam(width, height, FoVy, FoVx, znear, zfar, wv.contiguous(), full.contiguous(), center.contiguous())


def orbit_camera(k, n_views, width, height, radius=6.0, fovx_deg=60.0, target=(0.0, 0.0, 0.0), height_y=0.0):
    """k-th of n_views cameras on a circle of `radius` in the y=height_y plane, looking at
    `target`, up = -y (camera y axis points along +y world, as in COLMAP scenes)."""
    ang = 2.0 * math.pi * k / n_views
    c = np.array([radius * math.sin(ang), height_y, -radius * math.cos(ang)], dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - c
    f /= np.linalg.norm(f)
    down = np.array([0.0, 1.0, 0.0])
    x = np.cross(down, f)
    x /= np.linalg.norm(x)
    y = np.cross(f, x)
    R = np.stack([x, y, f], axis=1)  # camera-to-world, columns = camera axes
    T = -R.transpose() @ c
    FoVx = math.radians(fovx_deg)
    FoVy = focal2fov(fov2focal(FoVx, width), height)
    return make_camera(R, T, width, height, FoVx, FoVy)


@dataclass
class SyntheticGaussians:
    """Raw (pre-activation) parameters with GaussianModel's getter semantics."""
    _xyz: torch.Tensor
    _features_dc: torch.Tensor
    _features_rest: torch.Tensor
    _scaling: torch.Tensor
    _rotation: torch.Tensor
    _opacity: torch.Tensor
    active_sh_degree: int
    max_sh_degree: int

    # scene/gaussian_model.py:36-47 setup_functions(): the activations are attributes of the model
    scaling_activation = staticmethod(torch.exp)
    opacity_activation = staticmethod(torch.sigmoid)
    rotation_activation = staticmethod(torch.nn.functional.normalize)

    # scene/gaussian_model.py:98-118
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier=1):
        """scene/gaussian_model.py:29-33,120-123 (device-agnostic restatement)."""
        s = scaling_modifier * self.get_scaling
        q = self._rotation / self._rotation.norm(dim=1, keepdim=True)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                          2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                          2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
        L = Rm * s[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)

    def to(self, device):
        return SyntheticGaussians(self._xyz.to(device), self._features_dc.to(device), self._features_rest.to(device),
                                  self._scaling.to(device), self._rotation.to(device), self._opacity.to(device),
                                  self.active_sh_degree, self.max_sh_degree)

    def requires_grad_(self, flag=True):
        for t in (self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity):
            t.requires_grad_(flag)
        return self

    @property
    def num(self):
        return self._xyz.shape[0]


def make_gaussians(N, sh_degree=3, seed=SEED, extent=(4.0, 2.25, 4.0), log_scale_mean=math.log(0.004),
                   log_scale_std=0.5, opacity_mean=-1.0, opacity_std=1.5, rest_std=0.05):
    """SURVEY.md section 8d frozen generator."""
    g = torch.Generator("cpu").manual_seed(seed)
    ext = torch.tensor(extent)
    xyz = (torch.rand(N, 3, generator=g) * 2 - 1) * ext
    scaling = torch.randn(N, 3, generator=g) * log_scale_std + log_scale_mean
    rotation = torch.randn(N, 4, generator=g)
    opacity = torch.randn(N, 1, generator=g) * opacity_std + opacity_mean
    M = (sh_degree + 1) ** 2
    dc = RGB2SH(torch.rand(N, 1, 3, generator=g))
    rest = torch.randn(N, M - 1, 3, generator=g) * rest_std
    return SyntheticGaussians(xyz, dc, rest, scaling, rotation, opacity, sh_degree, sh_degree)


def make_heavy_tailed(g, frac=0.02, radius=0.15, log_scale_mean=math.log(0.03), opacity_mean=-4.5, seed=SEED + 7, centre=(0.0, 0.0, 0.0)):
    """Turns a fraction of the Gaussians of `g` (in place, raw parameters) into what real captures have and the uniform
    generator lacks: a dense pile of larger, faint splats around one point every orbit camera looks at -- per-tile lists of
    tens of thousands of entries (uniform scene at C3: mean 507, max ~700), most of whose entries are rejected or contribute
    little, so lists are long AND early termination does not cut them short.  At C3 (3M, 1080p): frac 0.02 -> 60 k splats of
    ~8 px sigma over ~100 tiles: ~1 M extra instances, 20-25 k entries on the densest tiles (frac 0.04, measured: 47 955)."""
    n = int(g.num * frac)
    if n <= 0:
        return g
    gen = torch.Generator("cpu").manual_seed(seed)
    idx = torch.randperm(g.num, generator=gen)[:n]
    d = torch.randn(n, 3, generator=gen)
    d = d / d.norm(dim=1, keepdim=True) * radius * torch.rand(n, 1, generator=gen) ** (1.0 / 3.0)
    with torch.no_grad():
        g._xyz[idx] = d + torch.tensor(centre)
        g._scaling[idx] = torch.randn(n, 3, generator=gen) * 0.35 + log_scale_mean
        g._opacity[idx] = torch.randn(n, 1, generator=gen) * 0.5 + opacity_mean
    return g


def morton_permutation(xyz, bits=10):
    """Permutation that puts points into Morton (Z-curve) order of their positions: neighbours in space become neighbours in
    memory, for every camera.  The rasterizer does not need it -- results do not depend on the order of the Gaussians -- but the
    blend kernels gather 48-byte records by Gaussian id and the radix passes scatter keys by tile: with a spatially coherent
    order both touch far fewer distinct cache lines (bench.py --spatial-order)."""
    p = xyz.detach().float().cpu()
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).round().to(torch.int64)
    code = torch.zeros(p.shape[0], dtype=torch.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, stable=True)


def permute_(g, perm):
    """Reorders the raw parameter tensors of a SyntheticGaussians / GaussianModel-like object in place (no optimizer state)."""
    with torch.no_grad():
        for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            t = getattr(g, name)
            t.data = t.data[perm.to(t.device)].contiguous()
    return g


@dataclass
class PipelineParams:
    """arguments/__init__.py:72-77"""
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False
