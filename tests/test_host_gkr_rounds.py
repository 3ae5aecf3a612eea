"""The vectorised interaction-variable rounds of LogUp-GKR on the host (sp1_amd/csrc/gkr_host.cpp) against scalar extension
arithmetic from the oracle: sums and fold of one round on random tables, ragged pair counts included. Host only."""
import ctypes as C
import os

import numpy as np
import pytest

import pyoracle as orc

P = 0x7F000001
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sp1_amd", "lib", "libsp1hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("libsp1hip.so not built")
    h = C.CDLL(LIB)
    h.sp1hip_gkr_host_round_sums.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    h.sp1hip_gkr_host_round_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    if not h.sp1hip_gkr_host_simd_available():
        pytest.skip("no AVX-512 on this CPU: the library runs the scalar rounds")
    return h


def ext_add(a, b):
    return ((a.astype(np.uint64) + b) % P).astype(np.uint32)


def ext_sub(a, b):
    return ((a.astype(np.uint64) + P - b) % P).astype(np.uint32)


@pytest.mark.parametrize("real_pairs", [1, 2, 7, 8, 9, 16, 37, 365])
def test_round_sums_and_fold_match_scalar_arithmetic(lib, real_pairs):
    rng = np.random.default_rng(1000 + real_pairs)
    n = 2 * real_pairs
    stride = ((2 * ((real_pairs + 7) // 8 * 8) + 15) // 16) * 16 + 16
    # tab[(4 w + k) * stride + i]: garbage behind the live entries on purpose (the kernel must mask it)
    tab = rng.integers(0, P, size=(4, 4, stride), dtype=np.uint32)
    eq = rng.integers(0, P, size=(4, stride), dtype=np.uint32)
    ext = lambda t, i: np.ascontiguousarray(t[:, i])
    zero = np.zeros(4, np.uint32)
    want = [zero.copy() for _ in range(6)]
    for k in range(real_pairs):
        a, b = 2 * k, 2 * k + 1
        n0a, d0a, n1a, d1a = (ext(tab[w], a) for w in range(4))
        n0b, d0b, n1b, d1b = (ext(tab[w], b) for w in range(4))
        ea, eb = ext(eq, a), ext(eq, b)
        X = ext_add(orc.ext_mul(d0a, n1a), orc.ext_mul(d1a, n0a))
        Y = orc.ext_mul(d0a, d1a)
        want[0] = ext_add(want[0], orc.ext_mul(ea, X))
        want[1] = ext_add(want[1], orc.ext_mul(ea, Y))
        sn0, sn1, sd0, sd1, es = ext_add(n0a, n0b), ext_add(n1a, n1b), ext_add(d0a, d0b), ext_add(d1a, d1b), ext_add(ea, eb)
        Xh = ext_add(orc.ext_mul(sd0, sn1), orc.ext_mul(sd1, sn0))
        Yh = orc.ext_mul(sd0, sd1)
        want[2] = ext_add(want[2], orc.ext_mul(es, Xh))
        want[3] = ext_add(want[3], orc.ext_mul(es, Yh))
        want[4] = ext_add(want[4], ea)
        want[5] = ext_add(want[5], es)
    got = np.zeros((6, 4), np.uint32)
    assert lib.sp1hip_gkr_host_round_sums(tab.ctypes.data, stride, eq.ctypes.data, stride, real_pairs, got.ctypes.data) == 0
    assert np.array_equal(got, np.stack(want))

    alpha = rng.integers(0, P, size=4, dtype=np.uint32)
    out = np.zeros_like(tab)
    assert lib.sp1hip_gkr_host_round_fold(tab.ctypes.data, out.ctypes.data, stride, real_pairs, alpha.ctypes.data) == 0
    for w in range(4):
        for k in range(real_pairs):
            lo, hi = ext(tab[w], 2 * k), ext(tab[w], 2 * k + 1)
            assert np.array_equal(ext(out[w], k), ext_add(lo, orc.ext_mul(alpha, ext_sub(hi, lo)))), (w, k)
    assert n <= stride
