"""CPU checks of the oracle's jagged PCS evaluation proof: prover -> restated reference
verifier round trips over ragged tables (zero-row tables, odd heights, full-height tables, one or two
commitment rounds), soundness negatives, and prover-table vs verifier-branching-program consistency. The
verifier used here is the one that accepts the reference's real proof (test_oracle_golden.py)."""
import numpy as np
import pytest

import pyoracle as orc

P = 0x7F000001


def make_rounds(shapes_per_round, L, lsh, batch, seed, lb=1):
    rounds, tables_per_round = [], []
    for r, shapes in enumerate(shapes_per_round):
        tabs = [orc.random_felts((h, w), seed + 100 * r + k) if h else np.zeros((0, w), np.uint32)
                for k, (h, w) in enumerate(shapes)]
        tables_per_round.append(tabs)
        rounds.append(orc.JaggedRound(tabs, L, lsh, batch, lb))
    return rounds, tables_per_round


def claims_for(tables_per_round, L, z_row):
    out = []
    for tabs in tables_per_round:
        cl = [orc.padded_column_openings(t, L, z_row) for t in tabs]
        out.append(np.concatenate(cl) if cl else np.zeros((0, 4), np.uint32))
    return out


CASES = [
    ([[(8, 3), (5, 2)]], 3, 2, 2),                                   # one round, odd height
    ([[(16, 2), (0, 3), (7, 1)], [(16, 4), (1, 2), (9, 3)]], 4, 3, 2),   # two rounds, a zero-row table, single row
    ([[(4, 1)], [(32, 5), (31, 2), (2, 7)]], 5, 3, 4),               # full-height table, wide batches
    ([[(6, 2), (6, 2), (0, 1), (0, 4)]], 3, 4, 3),                   # stacking height > table heights, empty tail tables
]


@pytest.mark.parametrize("shapes,L,lsh,batch", CASES)
def test_jagged_roundtrip(shapes, L, lsh, batch):
    rounds, tabs = make_rounds(shapes, L, lsh, batch, 7 + L)
    ch = orc.Challenger()
    for r in rounds:
        ch.observe(r.commit)
    z_row = ch.sample_point(L)
    claims = claims_for(tabs, L, z_row)
    v = ch.clone()
    blob = orc.jagged_prove(z_row, claims, rounds, lsh, ch, 1, 6, 4)
    commits = [r.commit for r in rounds]
    end = v.clone()
    assert orc.jagged_verify(commits, z_row, claims, blob, lsh, end, 1, 6, 4) == 0
    assert np.array_equal(end.state(), ch.state())               # prover and verifier transcripts end equal
    # negatives: a flipped proof byte anywhere in the jagged part, a wrong claim, a wrong commitment
    for off in (len(blob) - 30, len(blob) - 150, len(blob) // 2):
        bad = bytearray(blob)
        bad[off] ^= 1
        assert orc.jagged_verify(commits, z_row, claims, bytes(bad), lsh, v.clone(), 1, 6, 4) != 0
    wrong = [c.copy() for c in claims]
    wrong[-1][0, 0] = (int(wrong[-1][0, 0]) + 1) % P
    assert orc.jagged_verify(commits, z_row, wrong, blob, lsh, v.clone(), 1, 6, 4) != 0
    wc = [c.copy() for c in commits]
    wc[0][3] ^= 1
    assert orc.jagged_verify(wc, z_row, claims, blob, lsh, v.clone(), 1, 6, 4) != 0


def test_jagged_table_agrees_with_branching_program():
    """The prover's materialised J table, evaluated as a multilinear at a random z_index, equals the
    verifier's branching-program evaluation (poly.rs test_single_table_jagged_eval generalised to ragged
    columns)."""
    heights = [5, 8, 0, 3, 8, 1]
    L = 3
    ch = orc.Challenger()
    ch.observe(orc.random_felts((5,), 3))
    z_row, z_col = ch.sample_point(L), ch.sample_point(3)
    table = orc.partial_jagged_table(heights, L, z_row, z_col)
    log_m = table.shape[0].bit_length() - 1
    z_index = ch.sample_point(log_m)
    eq = orc.partial_lagrange(z_index)
    import kb_py
    acc = [0, 0, 0, 0]
    for e, v in zip(orc.from_monty(eq), orc.from_monty(table)):
        acc = kb_py.ext_add(acc, kb_py.ext_mul([int(x) for x in e], [int(x) for x in v]))
    want = orc.from_monty(orc.full_jagged_eval(heights, z_row, z_col, z_index))
    assert acc == [int(x) for x in want]
    # on the hypercube J is the indicator: dense index 6 is column 1 (prefix 5), row 1
    def bits(x, n):
        return orc.to_monty(np.array([[(x >> (n - 1 - i)) & 1, 0, 0, 0] for i in range(n)], np.uint32))
    one = orc.full_jagged_eval(heights, bits(1, L), bits(1, 3), bits(6, log_m))
    assert list(orc.from_monty(one)) == [1, 0, 0, 0]
    zero = orc.full_jagged_eval(heights, bits(2, L), bits(1, 3), bits(6, log_m))
    assert list(orc.from_monty(zero)) == [0, 0, 0, 0]
