"""Real spherical harmonics (degrees 0-3) in PyTorch -- the `pipe.convert_SHs_python` alternate of render()
(reference: utils/sh_utils.py:57-120 eval_sh, :123-128 RGB2SH / SH2RGB).

Table-driven: every basis function is a constant times up to two direction factors, and the colour is the running sum
    result <- result + ((k * f1) * f2) * sh[..., i]          i = 0 .. (deg+1)^2 - 1
evaluated in coefficient order.  That is the same sequence of float32 operations as the reference's expression, so the
result is bit-identical to its eval_sh (tests/test_golden_reference_python.py compares against outputs of the reference
function itself), while the 16 basis terms live in one table instead of one long expression per degree.
"""

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def _basis_table(x, y, z):
    """[(constant, factor1 | None, factor2 | None)] for coefficients 1..15; factors are tensors broadcastable to the
    colour.  The association (k*f1)*f2 and the polynomial forms follow utils/sh_utils.py:79-101 exactly."""
    xx, yy, zz = x * x, y * y, z * z
    xy, yz, xz = x * y, y * z, x * z
    return [
        (-SH_C1, y, None), (SH_C1, z, None), (-SH_C1, x, None),
        (SH_C2[0], xy, None), (SH_C2[1], yz, None), (SH_C2[2], 2.0 * zz - xx - yy, None), (SH_C2[3], xz, None), (SH_C2[4], xx - yy, None),
        (SH_C3[0], y, 3 * xx - yy), (SH_C3[1], xy, z), (SH_C3[2], y, 4 * zz - xx - yy), (SH_C3[3], z, 2 * zz - 3 * xx - 3 * yy),
        (SH_C3[4], x, 4 * zz - xx - yy), (SH_C3[5], z, xx - yy), (SH_C3[6], x, xx - 3 * yy),
    ]


def eval_sh(deg, sh, dirs):
    """sh: [..., C, >= (deg+1)^2] coefficients, dirs: [..., 3] unit vectors -> [..., C] (before the +0.5 / clamp of render)."""
    if not 0 <= deg <= 3:
        raise AssertionError("SH degree must be 0..3")
    n = (deg + 1) ** 2
    if sh.shape[-1] < n:
        raise AssertionError("not enough SH coefficients for this degree")
    result = SH_C0 * sh[..., 0]
    if deg == 0:
        return result
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    for i, (k, f1, f2) in enumerate(_basis_table(x, y, z)[:n - 1], start=1):
        w = k * f1
        if f2 is not None:
            w = w * f2
        result = result + w * sh[..., i]
    return result


def RGB2SH(rgb):
    """colour -> degree-0 coefficient (utils/sh_utils.py:123-124)"""
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh):
    """degree-0 coefficient -> colour (utils/sh_utils.py:127-128)"""
    return sh * SH_C0 + 0.5
