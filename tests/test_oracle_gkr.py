"""CPU checks of the oracle's LogUp-GKR: dense prover -> restated reference verifier
(including the final interaction check against the opened trace values and the zero cumulative sum)."""
import numpy as np
import pytest

import pyoracle as orc
from gkr_chips import make_gkr_chips

P = 0x7F000001


@pytest.mark.parametrize("n_tuples,L,with_empty,dup", [
    (4, 3, False, 2),        # Alpha fills 2^L rows exactly
    (5, 4, True, 3),         # odd heights, an empty chip, 15 of 16 rows
    (1, 1, False, 2),        # a single row variable: no GKR rounds at all
    (3, 5, False, 1),        # mostly padding
])
def test_gkr_roundtrip(n_tuples, L, with_empty, dup):
    chips = make_gkr_chips(n_tuples, 10 + L, with_empty, dup)
    assert all(c[1].shape[0] <= 1 << L for c in chips)
    ch = orc.Challenger()
    ch.observe(orc.random_felts((9,), L))
    v = ch.clone()
    blob = orc.gkr_prove(chips, L, ch)
    heights = [c[1].shape[0] for c in chips]
    end = v.clone()
    assert orc.gkr_verify(chips, heights, L, blob, end) == 0
    assert np.array_equal(end.state(), ch.state())
    for off in (40, len(blob) // 2, len(blob) - 40):
        bad = bytearray(blob)
        bad[off] ^= 1
        assert orc.gkr_verify(chips, heights, L, bytes(bad), v.clone()) != 0
    if L > 1:
        wrong = list(heights)
        wrong[0] -= 1                                # a wrong height changes the geq correction
        assert orc.gkr_verify(chips, wrong, L, blob, v.clone()) != 0


def test_gkr_rejects_unbalanced_interactions():
    chips = make_gkr_chips(4, 3, False, 2)
    chips[2][1][0, 0] = (int(chips[2][1][0, 0]) + 1) % P      # Gamma receives a tuple nobody sent
    ch = orc.Challenger()
    v = ch.clone()
    blob = orc.gkr_prove(chips, 3, ch)
    assert orc.gkr_verify(chips, [c[1].shape[0] for c in chips], 3, blob, v) == 4     # cumulative sum != 0


@pytest.mark.parametrize("n_tuples,L,with_empty,dup", [
    (4, 3, False, 2), (5, 4, True, 3), (1, 1, False, 2), (2, 2, False, 1), (3, 5, False, 1), (9, 6, True, 2), (37, 7, True, 3),
    (300, 10, True, 3),
])
def test_jagged_aware_prover_equals_the_dense_one(n_tuples, L, with_empty, dup):
    """gkr_prove_sparse (real rows + closed-form padding: the reference's CPU shape, what bench.py's cpu_baseline times)
    writes the same bytes and leaves the same transcript as the dense formulation."""
    chips = make_gkr_chips(n_tuples, 10 + L, with_empty, dup)
    a, b = orc.Challenger(), orc.Challenger()
    seed = orc.random_felts((9,), L)
    a.observe(seed)
    b.observe(seed)
    dense = orc.gkr_prove(chips, L, a)
    try:
        orc.set_gkr_sparse(True)
        sparse = orc.gkr_prove(chips, L, b)
    finally:
        orc.set_gkr_sparse(False)
    assert sparse == dense
    assert np.array_equal(a.state(), b.state())
