"""CPU checks of the oracle's whole shard proof: prove_shard_with_data -> restated verify_shard with every
chip-dependent check (interactions, constraints, row/column counts), soundness negatives."""
import numpy as np
import pytest

import pyoracle as orc
from shard_chips import make_shard_chips, preprocessed_round

LB, NQ, PW = 1, 5, 4


@pytest.mark.parametrize("n_tuples,L,lsh,batch,with_empty,dup", [
    (4, 3, 2, 2, False, 2),
    (5, 4, 3, 3, True, 3),
    (3, 5, 2, 4, False, 1),
])
def test_shard_roundtrip(n_tuples, L, lsh, batch, with_empty, dup):
    chips, publics = make_shard_chips(n_tuples, 20 + L, with_empty, dup)
    prep = preprocessed_round(chips, L, lsh, batch, LB)
    ch = orc.Challenger()
    ch.observe(prep.commit)                                # stands for vk.observe_into
    v = ch.clone()
    blob = orc.shard_prove(chips, publics, prep, L, lsh, batch, ch, LB, NQ, PW)
    end = v.clone()
    assert orc.shard_verify(chips, prep.commit, blob, L, lsh, end, LB, NQ, PW) == 0
    assert np.array_equal(end.state(), ch.state())
    for off in (4, 60, len(blob) // 3, len(blob) // 2, len(blob) - 100):
        bad = bytearray(blob)
        bad[off] ^= 1
        assert orc.shard_verify(chips, prep.commit, bytes(bad), L, lsh, v.clone(), LB, NQ, PW) != 0
    wrong = prep.commit.copy()
    wrong[0] ^= 1
    assert orc.shard_verify(chips, wrong, blob, L, lsh, v.clone(), LB, NQ, PW) != 0


def test_shard_rejects_a_violated_constraint():
    chips, publics = make_shard_chips(4, 9)
    chips[0][2][1, 2] = orc.to_monty(np.array([2], np.uint32))[0]      # Alpha: m = 2 breaks m (m - 1) = 0 (and the lookup balance)
    prep = preprocessed_round(chips, 3, 2, 2, LB)
    ch = orc.Challenger()
    v = ch.clone()
    blob = orc.shard_prove(chips, publics, prep, 3, 2, 2, ch, LB, NQ, PW)
    assert orc.shard_verify(chips, prep.commit, blob, 3, 2, v, LB, NQ, PW) != 0
