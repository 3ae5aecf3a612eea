"""The transcript's host permutation (sp1_amd/csrc/p2_host.cpp): the AVX-512 form, the scalar integer form and the scalar
fp64 form of the library against each other and against the oracle's Poseidon2, word for word. Host only: no GPU."""
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
    h.sp1hip_poseidon2_permute_host.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    h.sp1hip_poseidon2_permute_host.restype = C.c_int
    return h


def edge_states():
    rows = [np.zeros(16, np.uint32), np.full(16, P - 1, np.uint32), np.arange(16, dtype=np.uint32)]
    for k in range(16):                      # one extreme word at a time
        r = np.zeros(16, np.uint32); r[k] = P - 1; rows.append(r)
        r = np.full(16, P - 1, np.uint32); r[k] = 0; rows.append(r)
    return np.stack(rows)


def test_forms_agree_and_match_the_oracle(lib):
    rng = np.random.default_rng(20260924)
    states = np.concatenate([edge_states(), rng.integers(0, P, size=(20000, 16), dtype=np.uint32)])
    out = []
    for form in (0, 1, 2):
        s = states.copy()
        assert lib.sp1hip_poseidon2_permute_host(s.ctypes.data, len(s), form) == 0
        assert (s < P).all()
        out.append(s)
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[1], out[2])
    for i in list(range(40)) + list(range(len(states) - 200, len(states))):
        assert np.array_equal(orc.permute(states[i]), out[0][i]), i


def test_chained_like_a_sponge(lib):
    """each permutation feeds the next one, 8 fresh rate words in between (what the duplex challenger does)"""
    rng = np.random.default_rng(7)
    a = np.zeros((1, 16), np.uint32)
    b = a.copy()
    for _ in range(300):
        rate = rng.integers(0, P, size=8, dtype=np.uint32)
        a[0, :8] = rate
        b[0, :8] = rate
        assert lib.sp1hip_poseidon2_permute_host(a.ctypes.data, 1, 0) == 0
        b[0] = orc.permute(b[0])
    assert np.array_equal(a, b)


def test_rejects_non_canonical_words(lib):
    s = np.zeros((1, 16), np.uint32)
    s[0, 3] = P
    assert lib.sp1hip_poseidon2_permute_host(s.ctypes.data, 1, 0) != 0
