"""CPU restatement of the segment-parallel forward walk of a long tile list (lg_blend_fwd_seg / _scan / _rewalk),
per pixel, in float32: free-running segment products, prefix scan with the parking rule, exact re-walk of the parked segment.
Against the sequential walk of the published algorithm it must give the SAME last contributor (n_contrib) and stop position,
and transmittance / colour to float rounding -- for lists that never terminate, terminate early, or sit near the threshold.
(The device kernels are compared with the serial walk and the oracle by tests/test_gpu_long_tiles.py on the GPU.)"""
import numpy as np

F = np.float32
T_MIN = F(1e-4)
A_MIN = F(1.0 / 255.0)


def sequential(alpha, col):
    T, C, last = F(1.0), F(0.0), 0
    for i, (a, c) in enumerate(zip(alpha, col)):
        if a < A_MIN:
            continue
        tt = T * (F(1.0) - a)
        if tt < T_MIN:
            break
        C = C + c * (a * T)
        T = tt
        last = i + 1
    return T, C, last


def walk_exact(alpha, col, lo, hi, T):
    """the pair step of fwd_pair over entries [lo, hi) from transmittance T: (T, colour inside, last, stopped)"""
    C, last = F(0.0), 0
    for i in range(lo, hi):
        a = alpha[i]
        if a < A_MIN:
            continue
        tt = T * (F(1.0) - a)
        if tt < T_MIN:
            return T, C, last, True
        C = C + col[i] * (a * T)
        T = tt
        last = i + 1
    return T, C, last, False


def parallel(alpha, col, S):
    n = len(alpha)
    nseg = (n + S - 1) // S
    seg = []
    for s in range(nseg):                               # pass 1: free-running, T_local from 1, no termination
        T, C, last = F(1.0), F(0.0), 0
        for i in range(s * S, min(n, s * S + S)):
            a = alpha[i]
            if a < A_MIN:
                continue
            C = C + col[i] * (a * T)
            T = T * (F(1.0) - a)
            last = i + 1
        seg.append((T, C, last))
    T, C, last, sstar = F(1.0), F(0.0), 0, None          # pass 2: scan, park where T P_s comes near the threshold
    for s, (P, Cs, ls) in enumerate(seg):
        Tend = T * P
        if not (Tend >= T_MIN * F(1.001)):
            sstar = s
            break
        C = C + T * Cs
        T = Tend
        last = ls if ls else last
    if sstar is None:
        return T, C, last
    cur, stopped = sstar, False                          # pass 3: exact re-walk from the parked segment on
    while cur < nseg and not stopped:
        T, Cin, ls, stopped = walk_exact(alpha, col, cur * S, min(n, cur * S + S), T)
        C = C + Cin
        last = ls if ls else last
        cur += 1
    return T, C, last


def test_parallel_walk_equals_the_sequential_one():
    rs = np.random.RandomState(4)
    for trial in range(400):
        n = int(rs.choice([70, 200, 700, 3000]))
        S = int(rs.choice([64, 128, 1024]))
        kind = trial % 4
        if kind == 0:      # faint pile: never terminates
            alpha = rs.uniform(0.0, 0.012, n)
        elif kind == 1:    # semi-opaque: terminates somewhere in the middle
            alpha = rs.uniform(0.0, 0.08, n)
        elif kind == 2:    # opaque: terminates inside the first segment
            alpha = rs.uniform(0.2, 0.99, n)
        else:              # engineered to pass the threshold region slowly (many products near 1e-4)
            alpha = np.concatenate([rs.uniform(0.05, 0.1, 120), rs.uniform(0.004, 0.006, n - 120)]) if n > 120 else rs.uniform(0.05, 0.1, n)
        alpha = alpha.astype(F)
        col = rs.uniform(0, 1, n).astype(F)
        Ts, Cs, ls = sequential(alpha, col)
        Tp, Cp, lp = parallel(alpha, col, S)
        assert lp == ls, (trial, n, S, kind, ls, lp)
        assert abs(float(Tp) - float(Ts)) <= 1e-5 * max(float(Ts), 1e-4) + 1e-9, (trial, Ts, Tp)   # regrouped products of up to 3000 factors
        assert abs(float(Cp) - float(Cs)) <= 3e-6, (trial, Cs, Cp)
