"""CPU check of the algorithm behind lg_block_hit (lightgaussian_amd/csrc/lg_blend.h): a float32 numpy restatement
of its two clamped 1-D maximisations, tested for CONSERVATIVENESS against brute force over the 64 pixels of the block --
whenever the test drops a block, no pixel of it may pass alpha >= 1/255.  (The device code itself is exercised by the -m gpu
parity tests and tools/gpu_fuzz.py; this pins the geometry and the margin.)"""
import numpy as np

F = np.float32


def block_reach(cx, cy, ha, nb, hc, op, x0, y0):
    """float32, operation for operation as lg_reach / lg_block_hit (without the box test): True = keep the block."""
    tau = F(np.log(F(255.0) * op)) + F(1.0e-3)
    r2ha, r2hc = F(1.0) / (F(2.0) * ha), F(1.0) / (F(2.0) * hc)
    x1, y1 = x0 + F(7.0), y0 + F(7.0)
    dxe = np.minimum(np.maximum(cx, x0), x1) - cx
    dye = np.minimum(np.maximum(cy, y0), y1) - cy
    dy1 = np.minimum(np.maximum(cy - nb * dxe * r2hc, y0), y1) - cy
    p1 = (ha * dxe + nb * dy1) * dxe + hc * dy1 * dy1
    dx2 = np.minimum(np.maximum(cx - nb * dye * r2ha, x0), x1) - cx
    p2 = (ha * dx2 + nb * dye) * dx2 + hc * dye * dye
    return np.maximum(p1, p2) >= -tau


def test_dropped_blocks_hold_no_contributing_pixel():
    rs = np.random.RandomState(12)
    n = 200000
    # covariance with the +0.3 low-pass, random orientation and anisotropy up to ~10: the conics K1 lets through with culling on
    s1 = np.exp(rs.uniform(np.log(0.6), np.log(40.0), n)); s2 = s1 * np.exp(-rs.uniform(0.0, 2.3, n))
    th = rs.uniform(0, np.pi, n)
    a = (s1 * np.cos(th)) ** 2 + (s2 * np.sin(th)) ** 2 + 0.3
    c = (s1 * np.sin(th)) ** 2 + (s2 * np.cos(th)) ** 2 + 0.3
    b = (s1 ** 2 - s2 ** 2) * np.sin(th) * np.cos(th)
    det = a * c - b * b
    A, B, C = c / det, -b / det, a / det
    ha, nb, hc = (-0.5 * A).astype(F), (-B).astype(F), (-0.5 * C).astype(F)
    op = np.exp(rs.uniform(np.log(1.0 / 250.0), 0.0, n)).astype(F)
    x0 = (8 * rs.randint(0, 40, n)).astype(F); y0 = (8 * rs.randint(0, 30, n)).astype(F)
    # centres from well inside to far outside the block, so that both outcomes are common
    reach = np.sqrt(2.0 * np.log(255.0 * op.astype(np.float64)).clip(0) * np.maximum(a, c))
    cx = (x0 + 3.5 + rs.uniform(-1, 1, n) * (reach + 8)).astype(F); cy = (y0 + 3.5 + rs.uniform(-1, 1, n) * (reach + 8)).astype(F)
    keep = block_reach(cx, cy, ha, nb, hc, op, x0, y0)
    # brute force in float64 over the 64 pixel centres: does any pixel pass alpha >= 1/255 ?
    px = x0[:, None].astype(np.float64) + np.arange(8)[None, :]
    py = y0[:, None].astype(np.float64) + np.arange(8)[None, :]
    dx = cx[:, None, None].astype(np.float64) - px[:, None, :]
    dy = cy[:, None, None].astype(np.float64) - py[:, :, None]
    power = ha[:, None, None].astype(np.float64) * dx * dx + nb[:, None, None].astype(np.float64) * dx * dy + hc[:, None, None].astype(np.float64) * dy * dy
    alpha = np.minimum(0.99, op[:, None, None].astype(np.float64) * np.exp(power))
    contributes = ((power <= 0) & (alpha >= 1.0 / 255.0)).any(axis=(1, 2))
    assert 0.2 < keep.mean() < 0.9 and 0.1 < contributes.mean() < 0.8          # both outcomes well represented
    assert not (contributes & ~keep).any(), int((contributes & ~keep).sum())    # conservative: nothing that contributes is dropped
    # and tight: of the kept blocks most do hold a contributing pixel (the rest: the ellipse reaches the block between pixel centres)
    assert (contributes & keep).sum() / keep.sum() > 0.8


def test_centre_inside_the_block_always_keeps_it():
    rs = np.random.RandomState(3)
    n = 5000
    ha = -np.exp(rs.uniform(-6, 1, n)).astype(F); hc = -np.exp(rs.uniform(-6, 1, n)).astype(F)
    nb = (rs.uniform(-1.9, 1.9, n) * np.sqrt(ha.astype(np.float64) * hc.astype(np.float64))).astype(F)   # keeps the form negative definite
    x0 = (8 * rs.randint(0, 10, n)).astype(F); y0 = (8 * rs.randint(0, 10, n)).astype(F)
    cx = (x0 + rs.uniform(0, 7, n)).astype(F); cy = (y0 + rs.uniform(0, 7, n)).astype(F)
    op = np.full(n, 1.0 / 255.0, F)                                           # tau = 1e-3: the smallest reach there is
    assert block_reach(cx, cy, ha, nb, hc, op, x0, y0).all()
