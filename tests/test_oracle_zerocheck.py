"""CPU checks of the oracle's zerocheck (a9–a12): prover -> reference-verifier-equation round trips over
hand-written AIRs with ragged / odd / empty / full heights, soundness negatives, and the sumcheck round
consistency of the reference's REAL zerocheck proof (golden)."""
import os

import numpy as np
import pytest

import pyoracle as orc
from zc_airs import make_chips

P = 0x7F000001
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "kb_shrink_basefold.npz"))


def setup(heights, L, seed):
    pv = np.array([77, 12345], np.uint32)
    publics = orc.to_monty(pv)
    chips = make_chips(heights, seed, pv)
    ch = orc.Challenger()
    ch.observe(orc.random_felts((8,), seed))
    zeta = ch.sample_point(L)
    alpha, gkr = ch.sample_ext(), ch.sample_ext()
    zc = []
    for name, air, main, prep in chips:
        op = [orc.padded_column_openings(main, L, zeta)]
        if prep is not None:
            op.append(orc.padded_column_openings(prep, L, zeta))
        zc.append(orc.ZcChip(air.to_array(), air.main_width, air.prep_width, air.num_constraints, main, prep,
                             np.concatenate(op)))
    return chips, zc, zeta, alpha, gkr, publics, ch


@pytest.mark.parametrize("heights,L", [
    ({"Mul": 8}, 3),                                            # full height
    ({"Mul": 5, "Affine": 3, "Sbox": 6}, 3),                    # odd heights, nonzero padded-row adjustment, prep
    ({"Affine": 1, "Mul": 1}, 2),                               # single rows
    ({"Affine": 7, "Empty": 0, "Sbox": 2}, 4),                  # a pure-padding chip in the middle
    ({"Affine": 16, "Mul": 11, "Sbox": 16}, 4),
    ({"Mul": 1}, 1),                                            # one variable
    ({"Affine": 2}, 1),
    ({"Chain": 6, "Manyregs": 5, "Mul": 3}, 3),                 # big cones, > 32 live registers, untouched columns
])
def test_zerocheck_roundtrip(heights, L):
    chips, zc, zeta, alpha, gkr, publics, ch = setup(heights, L, 5 + L)
    vch = ch.clone()
    blob = orc.zerocheck_prove(zc, L, zeta, alpha, gkr, publics, ch)
    hs = [c.real_rows for c in zc]
    assert orc.zerocheck_verify(zc, hs, L, zeta, alpha, gkr, publics, blob, vch.clone()) == 0
    # a wrong height on a chip whose zero row violates its constraints is rejected
    for k, c in enumerate(zc):
        if c.main_width == 3 and c.prep_width == 0 and hs[k] < (1 << L):
            bad = list(hs)
            bad[k] += 1
            assert orc.zerocheck_verify(zc, bad, L, zeta, alpha, gkr, publics, blob, vch.clone()) != 0
    t = bytearray(blob)
    t[40] ^= 1
    assert orc.zerocheck_verify(zc, hs, L, zeta, alpha, gkr, publics, bytes(t), vch.clone()) != 0


def test_zerocheck_rejects_unsatisfied_trace():
    chips, zc, zeta, alpha, gkr, publics, ch = setup({"Mul": 6, "Affine": 4}, 3, 9)
    k = next(i for i, c in enumerate(zc) if c.main_width == 4)
    zc[k].main[2, 2] = (int(zc[k].main[2, 2]) + 1) % P            # break c = a*b on one row of Mul
    zc[k].openings = orc.padded_column_openings(zc[k].main, 3, zeta)
    vch = ch.clone()
    blob = orc.zerocheck_prove(zc, 3, zeta, alpha, gkr, publics, ch)
    assert orc.zerocheck_verify(zc, [c.real_rows for c in zc], 3, zeta, alpha, gkr, publics, blob, vch) != 0


def test_zerocheck_is_deterministic():
    def run():
        _, zc, zeta, alpha, gkr, publics, ch = setup({"Mul": 5, "Sbox": 8}, 3, 1)
        return orc.zerocheck_prove(zc, 3, zeta, alpha, gkr, publics, ch)
    assert run() == run()


def test_golden_sumcheck_round_consistency():
    """The reference's real proof: each univariate message evaluated at the next challenge equals the
    next message's p(0) + p(1); the first message sums to claimed_sum; the last evaluates to the final
    eval. Pins coefficient order, eval_one_plus_eval_zero and the point order
    (proof.point = [alpha_last .. alpha_first])."""
    for name in ("zerocheck", "jagged_sumcheck", "jagged_eval"):
        polys = orc.to_monty(GOLD[name + "_polys"])
        rc = orc.sumcheck_rounds_consistent(polys, orc.to_monty(GOLD[name + "_claimed_sum"]),
                                            orc.to_monty(GOLD[name + "_point"]), orc.to_monty(GOLD[name + "_eval"]))
        assert rc == 0, (name, rc)
