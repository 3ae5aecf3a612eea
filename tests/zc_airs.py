"""Hand-written single-row AIRs (constraint programs) + satisfying traces for the zerocheck tests."""
import numpy as np

import pyoracle as orc
from sp1_amd.air import AirProgram

P = 0x7F000001


def air_mul():
    """c = a * b, d boolean. The zero row satisfies it (padded_row_adjustment = 0)."""
    p = AirProgram("Mul", 4)
    a, b, c, d = (p.main(i) for i in range(4))
    p.assert_eq(c, a * b)
    p.assert_zero(d * (d - 1))
    return p


def trace_mul(rows, rng):
    a = rng.integers(0, P, rows, dtype=np.uint64)
    b = rng.integers(0, P, rows, dtype=np.uint64)
    return np.stack([a, b, a * b % P, rng.integers(0, 2, rows, dtype=np.uint64)], axis=1).astype(np.uint32)


def air_affine():
    """y = x + 5 and z = 2 x + public[0]: NOT satisfied by the zero row (exercises padded_row_adjustment / geq)."""
    p = AirProgram("Affine", 3)
    x, y, z = (p.main(i) for i in range(3))
    p.assert_eq(y, x + 5)
    p.assert_eq(z, x * 2 + p.public(0))
    return p


def trace_affine(rows, rng, pv0):
    x = rng.integers(0, P, rows, dtype=np.uint64)
    return np.stack([x, (x + 5) % P, (2 * x + int(pv0)) % P], axis=1).astype(np.uint32)


def air_sbox():
    """Degree 3 with a preprocessed selector: s * (x^3 + k - y) = 0, w = x * x, plus a public-value term."""
    p = AirProgram("Sbox", 3, prep_width=2)
    x, y, w = (p.main(i) for i in range(3))
    s, k = p.prep(0), p.prep(1)
    p.assert_eq(w, x * x)
    p.assert_zero(s * (w * x + k - y))
    p.assert_zero(s * (s - 1))
    p.assert_zero((w - x * x) * p.public(1))
    return p


def trace_sbox(rows, rng):
    x = rng.integers(0, P, rows, dtype=np.uint64)
    k = rng.integers(0, P, rows, dtype=np.uint64)
    s = rng.integers(0, 2, rows, dtype=np.uint64)
    junk = rng.integers(0, P, rows, dtype=np.uint64)
    y = np.where(s == 1, (x * x % P * x + k) % P, junk)
    main = np.stack([x, y, x * x % P], axis=1).astype(np.uint32)
    prep = np.stack([s, k], axis=1).astype(np.uint32)
    return main, prep


def air_chain(n=60):
    """One assert whose cone (a 60-deep product/sum chain over 8 columns) is larger than the GPU chunk
    limit, plus columns 8..11 and a preprocessed column that no constraint reads (GKR TOUCH path)."""
    p = AirProgram("Chain", 12, prep_width=2)
    cols = [p.main(i) for i in range(8)]
    acc = cols[0]
    for k in range(n):
        acc = acc * cols[(k + 1) % 7] + cols[(k + 3) % 7] if k % 3 else acc + cols[k % 7] * 3
    p.assert_zero((acc - acc) + cols[7] * (cols[7] - 1))
    p.assert_zero(p.prep(0) * (p.prep(0) - 1))
    return p


def trace_chain(rows, rng):
    main = rng.integers(0, P, (rows, 12), dtype=np.uint64)
    main[:, 7] = rng.integers(0, 2, rows)
    prep = rng.integers(0, P, (rows, 2), dtype=np.uint64)
    prep[:, 0] = rng.integers(0, 2, rows)
    return main.astype(np.uint32), prep.astype(np.uint32)


def air_manyregs(n=48):
    """All n products are computed first and only then summed in reverse order: > 32 values are live at
    once, which forces the scratch-register tier of the GPU interpreter."""
    p = AirProgram("Manyregs", 4)
    a, b, c, d = (p.main(i) for i in range(4))
    prods = [(a + k) * (b + 2 * k + 1) for k in range(n)]
    tot = prods[-1]
    for q in reversed(prods[:-1]):
        tot = tot + q
    # sum_k (a + k)(b + 2k + 1) = n a b + a S1 + b S0 + S2  with S0 = sum k, S1 = sum (2k+1), S2 = sum k (2k+1)
    s0 = sum(range(n)); s1 = sum(2 * k + 1 for k in range(n)); s2 = sum(k * (2 * k + 1) for k in range(n))
    p.assert_eq(c, tot)
    p.assert_eq(d, a * b)
    p.assert_eq(c, d * n + a * s1 + b * s0 + s2)
    return p


def trace_manyregs(rows, rng, n=48):
    a = rng.integers(0, P, rows, dtype=np.uint64)
    b = rng.integers(0, P, rows, dtype=np.uint64)
    s0 = sum(range(n)); s1 = sum(2 * k + 1 for k in range(n)); s2 = sum(k * (2 * k + 1) for k in range(n))
    ab = a * b % P
    c = (ab * n + a * s1 + b * s0 + s2) % P
    return np.stack([a, b, c, ab], axis=1).astype(np.uint32)


def make_chips(heights, seed, pv):
    """heights: dict name -> real rows; the name's prefix picks the AIR ('Affine*', 'Sbox*', anything else =
    Mul). Returns [(name, AirProgram, main, prep)] in BTreeMap (name) order like the reference, traces in
    Montgomery form, row-major, real rows only."""
    rng = np.random.default_rng(seed)
    out = []
    for name in sorted(heights):
        h = heights[name]
        if name.startswith("Affine"):
            air, main, prep = air_affine(), trace_affine(h, rng, pv[0]), None
        elif name.startswith("Sbox"):
            air = air_sbox()
            main, prep = trace_sbox(h, rng)
        elif name.startswith("Chain"):
            air = air_chain()
            main, prep = trace_chain(h, rng)
        elif name.startswith("Manyregs"):
            air, main, prep = air_manyregs(), trace_manyregs(h, rng), None
        else:
            air, main, prep = air_mul(), trace_mul(h, rng), None
        main_m = orc.to_monty(main) if h else np.zeros((0, air.main_width), np.uint32)
        prep_m = None if prep is None else (orc.to_monty(prep) if h else np.zeros((0, air.prep_width), np.uint32))
        out.append((name, air, main_m, prep_m))
    return out
