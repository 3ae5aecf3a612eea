"""Row-by-row checks of a machine description against concrete tables (numpy, canonical integers): every constraint
program evaluates to zero on every real row, and the interactions balance as a multiset (LogUp's claim). Independent
of both provers: it is how the trace generator and the hand transcription of the recursion machine are cross-checked
before any proof is made."""
import numpy as np

from sp1_amd.air import ADD, ASSERT_ZERO, CONST, LOAD_MAIN, LOAD_PREP, MUL, NEG, PUBLIC, SUB

P = 0x7F000001
U = np.uint64
PP = U(P)
R_INV = pow(1 << 32, -1, P)


def from_monty(x):
    return (np.asarray(x, dtype=U) * U(R_INV)) % PP


def constraint_values(air, prep, main, publics):
    """prep/main: canonical [rows, w] uint64; returns [rows, num_constraints] canonical."""
    rows = main.shape[0]
    vals, out = [], []
    for op, a, b in air.instrs:
        if op == LOAD_MAIN:
            v = main[:, a]
        elif op == LOAD_PREP:
            v = prep[:, a]
        elif op == CONST:
            v = np.full(rows, a, dtype=U)
        elif op == PUBLIC:
            v = np.full(rows, int(publics[a]), dtype=U)
        elif op == ADD:
            v = (vals[a] + vals[b]) % PP
        elif op == SUB:
            v = (vals[a] + PP - vals[b]) % PP
        elif op == MUL:
            v = (vals[a] * vals[b]) % PP
        elif op == NEG:
            v = (PP - vals[a]) % PP
        elif op == ASSERT_ZERO:
            out.append(vals[a])
            v = None
        else:                                   # HINT: no value, no constraint
            v = None
        vals.append(v)
    return np.stack(out, axis=1) if out else np.zeros((rows, 0), dtype=U)


def _vcol(v, prep, main):
    acc = np.full(main.shape[0], v.constant, dtype=U)
    for kind, col, w in v.terms:
        acc = (acc + (main if kind == "main" else prep)[:, col] * U(w)) % PP
    return acc


def bus_imbalance(chips):
    """chips: [(InteractionProgram, prep canonical, main canonical)]. Returns the messages whose signed multiplicity
    sum is non-zero mod p (empty dict = balanced)."""
    tally = {}
    for it, prep, main in chips:
        for sign, lst in ((1, it.sends), (-1, it.receives)):
            for kind, values, mult in lst:
                m = _vcol(mult, prep, main)
                cols = np.stack([_vcol(v, prep, main) for v in values], axis=1)
                live = np.nonzero(m)[0]
                for r in live:
                    key = (kind,) + tuple(int(x) for x in cols[r])
                    tally[key] = (tally.get(key, 0) + sign * int(m[r])) % P
    return {k: v for k, v in tally.items() if v}


def bus_imbalance_fast(chips, seed=1):
    """The same multiset check for large tables: every message is folded to a 64-bit key (random odd multipliers, wrapping
    arithmetic), keys are sorted and the signed multiplicities summed per key. Returns the number of keys whose sum is non-zero
    mod p (0 = balanced, up to a 2^-64-sized collision chance)."""
    rng = np.random.default_rng(seed)
    coef = rng.integers(0, 1 << 63, size=4097, dtype=np.int64).astype(U) * U(2) + U(1)
    keys, mults = [], []
    for it, prep, main in chips:
        for sign, lst in ((1, it.sends), (-1, it.receives)):
            for kind, values, mult in lst:
                m = _vcol(mult, prep, main)
                live = np.nonzero(m)[0]
                if not len(live):
                    continue
                mm, pl = main[live], (prep[live] if prep is not None else None)
                key = np.full(len(live), (kind + 1) * 0x9E3779B97F4A7C15 % (1 << 64), dtype=U)
                with np.errstate(over="ignore"):
                    for i, v in enumerate(values):
                        key = key * U(0x100000001B3) + _vcol(v, pl, mm) * coef[i]
                    key = key + U(len(values)) * coef[4096]
                ms = m[live].astype(np.int64)
                ms = np.where(ms > P // 2, ms - P, ms)              # multiplicities are small signed counts
                keys.append(key)
                mults.append(sign * ms)
    if not keys:
        return 0
    keys, mults = np.concatenate(keys), np.concatenate(mults)
    order = np.argsort(keys, kind="stable")
    ks, cs = keys[order], np.cumsum(mults[order])
    last = np.r_[ks[1:] != ks[:-1], True]
    ends = cs[last]
    sums = ends - np.r_[0, ends[:-1]]
    return int(np.count_nonzero(sums % P))


def check_shard(machine, tabs, publics, pv_program=None):
    """Every constraint of every chip on every row and every bus of one shard: returns (failing chips, unbalanced message keys).
    `pv_program`: the machine's eval_public_values as (AirProgram, InteractionProgram) — its constraints must vanish on `publics`
    and its sends / receives (the record's own: initial / final CPU state, the ends of the accumulation chains, range checks of
    the public limbs) join the buses; without it a RISC-V shard does NOT balance."""
    chips, bad = [], []
    pv = (publics.numpy() if hasattr(publics, "numpy") else np.asarray(publics)).astype(np.uint64)
    for air, it in machine:
        prep, main = tabs[air.name]
        m = main.cpu().numpy().astype(np.uint64)
        pr = prep.cpu().numpy().astype(np.uint64) if prep is not None else None
        assert (main >= 0).all() and (m < P).all(), air.name
        if air.num_constraints and m.shape[0] and constraint_values(air, pr, m, pv).any():
            bad.append(air.name)
        chips.append((it, pr, m))
    if pv_program is not None:
        air, it = pv_program
        row = pv[None, :it.main_width]
        if constraint_values(air, None, row, pv).any():
            bad.append(air.name)
        chips.append((it, None, row))
    return bad, bus_imbalance_fast(chips)


def check_exact(machine, tabs, publics, pv_program=None):
    """check_shard for small tables: ASSERTS that every constraint vanishes (naming the chip and the first failing constraint
    indices) and returns the exact tally of unbalanced messages ({} = balanced)."""
    chips = []
    pv = (publics.numpy() if hasattr(publics, "numpy") else np.asarray(publics)).astype(np.uint64)
    for air, it in machine:
        prep, main = tabs[air.name]
        m = main.cpu().numpy().astype(np.uint64)
        pr = prep.cpu().numpy().astype(np.uint64) if prep is not None else None
        assert (main >= 0).all() and (m < P).all(), air.name
        if air.num_constraints and m.shape[0]:
            cv = constraint_values(air, pr, m, pv)
            assert not cv.any(), (air.name, sorted(set(np.argwhere(cv != 0)[:, 1]))[:8])
        chips.append((it, pr, m))
    if pv_program is not None:
        air, it = pv_program
        row = pv[None, :it.main_width]
        cv = constraint_values(air, None, row, pv)
        assert not cv.any(), (air.name, sorted(set(np.argwhere(cv != 0)[:, 1]))[:8])
        chips.append((it, None, row))
    return bus_imbalance(chips)
