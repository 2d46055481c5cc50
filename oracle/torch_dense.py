"""Dense, autograd-differentiable restatement of the rasterizer in plain PyTorch (float64).

TEST INFRASTRUCTURE ONLY (same rules as oracle/lg_oracle.c).  Purpose: an *independent*
derivation of the gradients -- torch.autograd differentiates the forward formula, whereas
lg_oracle.c and the HIP kernels carry hand-derived backward passes.  O(P*N) memory: tiny
scenes only.  Reference-owned conventions it follows: see lg_oracle.c header.

Deliberate matches to the published backward's conventions (not "true" derivatives):
  * alpha = min(0.99, sigma*G) is differentiated straight-through,
  * the +-1.3*tan(fov) clamp of t.x/t.z, t.y/t.z passes no gradient,
  * scale_modifier is assumed 1.
The 1e-7 regulariser in 1/(det^2+1e-7) of the published cov2D backward is NOT reproduced
(relative effect <= 1.3e-5), so compare at ~1e-4.
"""

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """sh [N,M,3], dirs [N,3] -> [N,3]; same polynomial as utils/sh_utils.py:57-103."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
                       + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
                       + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def render_dense(*, means3D, means2D, opacities, W, H, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos,
                 sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
    dt = means3D.dtype
    N = means3D.shape[0]
    vm, pm = viewmatrix.to(dt), projmatrix.to(dt)
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    pview = ph @ vm  # row-vector convention
    phom = ph @ pm
    p_w = 1.0 / (phom[:, 3] + 1e-7)
    ndc = phom[:, :2] * p_w[:, None] + means2D[:, :2]
    tz = pview[:, 2]
    vis = tz > 0.2

    if cov3D_precomp is None:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                          2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                          2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(N, 3, 3)
        L = Rm * scales[:, None, :]
        Sig = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sig = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).view(N, 3, 3)

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = pview[:, 0] / tz, pview[:, 1] / tz
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz).detach(), pview[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz).detach(), pview[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).view(N, 2, 3)
    Wm = vm[:3, :3].t()  # Wm[c][k] = vm[k][c]
    T2 = J @ Wm
    cov = T2 @ Sig @ T2.transpose(1, 2)
    a, b, c_ = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c_ - b * b
    vis = vis & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    A, B, Cc = c_ / det_safe, -b / det_safe, a / det_safe
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    rad = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    ix = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    iy = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16

    def tr(v):
        return torch.trunc(v).to(torch.int64)
    ixd, iyd = ix.detach(), iy.detach()
    rx0 = tr((ixd - rad) / 16).clamp(0, gx); ry0 = tr((iyd - rad) / 16).clamp(0, gy)
    rx1 = tr((ixd + rad + 15) / 16).clamp(0, gx); ry1 = tr((iyd + rad + 15) / 16).clamp(0, gy)
    vis = vis & (((rx1 - rx0) * (ry1 - ry0)) > 0)
    radii = torch.where(vis, rad, torch.zeros_like(rad)).to(torch.int32)

    if colors_precomp is None:
        d = means3D - campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh(sh_degree, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp

    order = torch.argsort(tz.detach(), stable=True)
    py, px = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    px = px.reshape(-1); py = py.reshape(-1)
    tpx, tpy = px // 16, py // 16
    o = order
    in_rect = (tpx[:, None] >= rx0[o][None]) & (tpx[:, None] < rx1[o][None]) & \
              (tpy[:, None] >= ry0[o][None]) & (tpy[:, None] < ry1[o][None]) & vis[o][None]
    dx = ix[o][None] - px[:, None].to(dt)
    dy = iy[o][None] - py[:, None].to(dt)
    power = -0.5 * (A[o][None] * dx * dx + Cc[o][None] * dy * dy) - B[o][None] * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    a_raw = opacities.reshape(-1)[o][None] * G
    alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
    ok = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
    T_incl = torch.cumprod(1.0 - alpha, dim=1)
    terminated = T_incl < 0.0001
    live = ok & ~terminated
    alpha_l = torch.where(live, alpha, torch.zeros_like(alpha))
    T_incl_l = torch.cumprod(1.0 - alpha_l, dim=1)
    T_before = torch.cat([torch.ones(T_incl_l.shape[0], 1, dtype=dt), T_incl_l[:, :-1]], 1)
    w = alpha_l * T_before
    color = w @ rgb[o] + T_incl_l[:, -1:] * bg.to(dt)[None]
    count = torch.zeros(N, dtype=torch.int64)
    count[o] = live.sum(0)
    return color.t().reshape(3, H, W), radii, count
