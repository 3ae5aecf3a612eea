"""Chips with BOTH constraints (AirProgram) and balanced interactions (InteractionProgram) over the same traces,
for whole-shard-proof tests: the traces of tests/gkr_chips.py plus AIRs they satisfy."""
import numpy as np

import pyoracle as orc
from gkr_chips import make_gkr_chips
from sp1_amd.air import AirProgram

P = 0x7F000001


def make_shard_chips(n_tuples, seed, with_empty=False, dup=2, publics=(7, 11)):
    """[(AirProgram, InteractionProgram, main, prep or None)] in name order.
    Alpha (a, b, m):        m (m - 1) = 0 and m (a - a) = 0                 (zero row satisfies it)
    Beta  prep (a, b), main (m, s): (m - dup) = 0 — NOT satisfied by the zero row (padded-row adjustment),
                                    prep0 * (m - dup) = 0, degree 3 term s * m * (m - dup) = 0
    Gamma (s, one):         one (one - 1) = 0, (one - 1) * public[0] = 0
    Omega (x, y), no rows:  x y = 0"""
    gk = make_gkr_chips(n_tuples, seed, with_empty, dup)
    out = []
    for prog, main, prep in gk:
        if prog.name == "Alpha":
            air = AirProgram("Alpha", 3)
            a, b, m = (air.main(i) for i in range(3))
            air.assert_zero(m * (m - 1))
            air.assert_zero(m * (a - a) + b * 0)
        elif prog.name == "Beta":
            air = AirProgram("Beta", 2, prep_width=2)
            m, s = air.main(0), air.main(1)
            air.assert_zero(m - dup)
            air.assert_zero(air.prep(0) * (m - dup))
            air.assert_zero(s * m * (m - dup))
        elif prog.name == "Gamma":
            air = AirProgram("Gamma", 2)
            one = air.main(1)
            air.assert_zero(one * (one - 1))
            air.assert_zero((one - 1) * air.public(0))
        else:
            air = AirProgram(prog.name, 2)
            air.assert_zero(air.main(0) * air.main(1))
        out.append((air, prog, main, prep))
    return out, orc.to_monty(np.array(publics, np.uint32))


def preprocessed_round(chips, L, lsh, batch, log_blowup):
    """The proving key's preprocessed commitment round: the preprocessed traces of the chips that have one."""
    tabs = [c[3] for c in chips if c[3] is not None]
    return orc.JaggedRound(tabs, L, lsh, batch, log_blowup)
