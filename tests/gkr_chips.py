"""Hand-written chips with balanced LogUp interactions (every sent tuple is received with the same total
multiplicity, so the cumulative sum of a shard is zero) for the LogUp-GKR tests."""
import numpy as np

import pyoracle as orc
from sp1_amd.air import InteractionProgram, VCol

P = 0x7F000001


def make_gkr_chips(n_tuples, seed, with_empty=False, dup=2):
    """Returns [(InteractionProgram, main [rows][w] Montgomery row-major, prep or None)] in name order.

    * "Alpha"  (main a, b, m):         sends   kind 5 (a, 2a + 3b + 7) with multiplicity m
    * "Beta"   (prep a, b; main m, s): receives kind 5 (a, 2a + 3b + 7) with multiplicity m   [values from preprocessed columns]
                                       sends   kind 7 (s) with multiplicity 1
    * "Gamma"  (main s, one):          receives kind 7 (s) with multiplicity `one` (a column of ones)
    * "Omega"  zero rows, one send (only when with_empty)
    Alpha lists every tuple `dup` times with multiplicity 1; Beta lists it once with multiplicity dup."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, P, n_tuples, dtype=np.uint64)
    b = rng.integers(0, P, n_tuples, dtype=np.uint64)
    s = rng.integers(0, P, n_tuples, dtype=np.uint64)
    alpha = InteractionProgram("Alpha", 3)
    val = [VCol.main(0), VCol.main(0, 2) + VCol.main(1, 3) + 7]
    alpha.send(5, val, VCol.main(2))
    rows = np.concatenate([np.stack([a, b, np.ones_like(a)], axis=1)] * dup)
    alpha_main = rows[rng.permutation(len(rows))].astype(np.uint32)

    beta = InteractionProgram("Beta", 2, prep_width=2)
    beta.send(7, [VCol.main(1)], VCol.const(1))
    beta.receive(5, [VCol.prep(0), VCol.prep(0, 2) + VCol.prep(1, 3) + 7], VCol.main(0))
    beta_prep = np.stack([a, b], axis=1).astype(np.uint32)
    beta_main = np.stack([np.full_like(a, dup), s], axis=1).astype(np.uint32)

    gamma = InteractionProgram("Gamma", 2)
    gamma.receive(7, [VCol.main(0)], VCol.main(1))
    perm = rng.permutation(n_tuples)
    gamma_main = np.stack([s[perm], np.ones_like(s)], axis=1).astype(np.uint32)

    chips = [(alpha, orc.to_monty(alpha_main), None), (beta, orc.to_monty(beta_main), orc.to_monty(beta_prep)),
             (gamma, orc.to_monty(gamma_main), None)]
    if with_empty:
        omega = InteractionProgram("Omega", 2)
        omega.send(9, [VCol.main(0), VCol.main(1), VCol.const(3)], VCol.main(1))
        chips.append((omega, np.zeros((0, 2), np.uint32), None))
    return chips
