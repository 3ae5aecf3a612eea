"""The algebra behind the zerocheck's fused first two rounds (sp1_amd/csrc/zerocheck.hip, `zc_biv_*`; the reference's
sp1-gpu/crates/sys/include/zerocheck/bivariate.cuh:L1-L118): a chip's polynomial on the grid {0, 1, 2, 4}^2 — twelve node sums of
the constraints over row quads, four corner sums of the linear GKR term, one geq correction at the quad that holds the first padded
row — gives exactly the values the two SEQUENTIAL rounds compute (round 0 over row pairs, fold by the challenge, round 1 over pairs
of the folded table), for every table height. Pure field arithmetic in Python integers; the device side is checked byte for byte
against the oracle by the GPU tests."""
import random

import pytest

P = 0x7F000001
W = 3


def inv(a):
    return pow(a % P, P - 2, P)


def C(row):                 # a degree-3 constraint with a constant term (its value on a zero row is NOT zero: the padding case)
    a, b, c = row
    return (a * b * c + 5 * a * b + 7 * c + 11) % P


def g(row, gk):             # the linear GKR batching term
    return sum(x * k for x, k in zip(row, gk)) % P


def lerp(r0, r1, t):
    return [(x + t * (y - x)) % P for x, y in zip(r0, r1)]


def eq_tab(zs):             # eq over len(zs) variables; zs[-1] <-> bit 0 of the index
    tab = [1]
    for zv in zs:
        tab = [t * f % P for t in tab for f in ((1 - zv) % P, zv)]
    return tab


def run(nv, rows, rng):
    N = 1 << nv
    z = [rng.randrange(P) for _ in range(nv)]
    gk = [rng.randrange(P) for _ in range(W)]
    T = []
    for i in range(N):
        if i < rows:        # a real row satisfies the constraint
            a, b = rng.randrange(P), rng.randrange(P)
            T.append([a, b, (-(5 * a * b + 11)) * inv(a * b + 7) % P])
            assert C(T[-1]) == 0
        else:
            T.append([0, 0, 0])
    pad = C([0, 0, 0])
    geq = [1 if i >= rows else 0 for i in range(N)]

    def h_seq(tab, geqs, etab, t, eq_adj):
        s = 0
        for i in range(len(tab) // 2):
            r = lerp(tab[2 * i], tab[2 * i + 1], t)
            gq = (geqs[2 * i] + t * (geqs[2 * i + 1] - geqs[2 * i])) % P
            s += etab[i] * (C(r) + g(r, gk) - pad * gq)
        return s * eq_adj % P

    # ---- the sequential rounds
    h0 = {t: h_seq(T, geq, eq_tab(z[:nv - 1]), t, 1) for t in (0, 2, 4)}
    a0 = rng.randrange(P)
    T1 = [lerp(T[2 * i], T[2 * i + 1], a0) for i in range(N // 2)]
    geq1 = [(geq[2 * i] + a0 * (geq[2 * i + 1] - geq[2 * i])) % P for i in range(N // 2)]
    e2 = eq_tab(z[:nv - 2])
    adj = ((1 - z[nv - 1]) * (1 - a0) + z[nv - 1] * a0) % P
    h1 = {t: h_seq(T1, geq1, e2, t, adj) for t in (0, 2, 4)}

    # ---- the bivariate form
    XS = [0, 1, 2, 4]
    nq = (rows + 3) // 4

    def bil(r, x, y):       # r[2 x + y]
        return [(r00 + x * (r10 - r00) + y * (r01 - r00) + x * y * (r11 - r10 - r01 + r00)) % P for r00, r01, r10, r11 in zip(*r)]

    S = {(x, y): sum(e2[q] * C(bil(T[4 * q:4 * q + 4], x, y)) for q in range(nq)) % P
         for x in XS for y in XS if not (x in (0, 1) and y in (0, 1))}
    G = {(x, y): sum(e2[q] * g(T[4 * q + 2 * x + y], gk) for q in range(nq)) % P for x in (0, 1) for y in (0, 1)}
    qb, m = rows // 4, rows % 4

    def H(x, y):
        if x in (0, 1) and y in (0, 1):
            return G[x, y]
        gb = (G[0, 0] + x * (G[1, 0] - G[0, 0]) + y * (G[0, 1] - G[0, 0]) + x * y * (G[1, 1] - G[1, 0] - G[0, 1] + G[0, 0])) % P
        corr = 0
        if m:               # the quad of the first padded row also holds real rows
            i01, i10 = int(1 >= m), int(2 >= m)
            corr = pad * e2[qb] * (x * i10 + y * i01 + x * y * (1 - i10 - i01))
        return (S[x, y] + gb - corr) % P

    zX = z[nv - 2]
    h0b = {t: ((1 - zX) * H(0, t) + zX * H(1, t)) % P for t in (0, 2, 4)}

    def interp4(vals, xq):
        s = 0
        for k, xk in enumerate(XS):
            num, den = 1, 1
            for j, xj in enumerate(XS):
                if j != k:
                    num, den = num * (xq - xj) % P, den * (xk - xj) % P
            s += vals[k] * num * inv(den)
        return s % P

    h1b = {t: adj * interp4([H(t, y) for y in XS], a0) % P for t in (0, 2, 4)}
    assert h0 == h0b and h1 == h1b, (nv, rows)


@pytest.mark.parametrize("nv", [2, 3, 4])
def test_bivariate_grid_reproduces_the_two_sequential_rounds(nv):
    rng = random.Random(5 + nv)
    for rows in range(1, (1 << nv) + 1):
        run(nv, rows, rng)


def test_the_library_interpolation_at_every_node_and_at_the_extremes():
    """`zc_biv_interp` (what the bivariate kernels run per column and node: one 36-bit accumulation, one reduction) against plain
    integers — on Montgomery words, which the affine map preserves — incl. the all-(p - 1) and alternating extremes."""
    import ctypes as C

    import numpy as np

    from sp1_amd import _lib
    lib = _lib.load()
    NODES = [(0, 2), (0, 4), (1, 2), (1, 4), (2, 0), (2, 1), (2, 2), (2, 4), (4, 0), (4, 1), (4, 2), (4, 4)]
    rng = random.Random(11)
    cases = [(P - 1,) * 4, (0, P - 1, P - 1, 0), (P - 1, 0, 0, P - 1), (P - 1, 0, 0, 0), (0, 0, 0, P - 1), (0, P - 1, 0, 0), (1, 0, 0, 0)]
    cases += [tuple(rng.randrange(P) for _ in range(4)) for _ in range(200)]
    out = np.zeros(1, np.uint32)
    for r00, r01, r10, r11 in cases:
        for e, (x, y) in enumerate(NODES):
            _lib.check(lib.sp1hip_zerocheck_biv_interp_host(r00, r01, r10, r11, e, out.ctypes.data_as(_lib.u32p)))
            want = (r00 + x * (r10 - r00) + y * (r01 - r00) + x * y * (r11 - r10 - r01 + r00)) % P
            assert int(out[0]) == want, (r00, r01, r10, r11, e)
    with pytest.raises(_lib.Sp1HipError):
        _lib.check(lib.sp1hip_zerocheck_biv_interp_host(0, 0, 0, 0, 12, out.ctypes.data_as(_lib.u32p)))
