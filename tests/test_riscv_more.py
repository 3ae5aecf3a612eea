"""The chips real programs' shards contain beyond the 30 of riscv.py (sp1_amd/machines/riscv_more.py): DivRem,
SyscallCore / SyscallPrecompile / SyscallInstrs, MemoryGlobalInit / Finalize, KeccakPermute and its controller.

Pins, as for riscv.py: column counts == rv64im_costs.json, constraint counts == rv64im_complexity.json, interaction counts == the
recorded core shard where it has the chip (DivRem 135, SyscallCore 4, SyscallInstrs 30) — and semantics: executed traces
(riscv_trace.py: DIV/REM and ECALL instructions in the loop body; riscv_more_trace.py: a Keccak precompile shard whose
permutation is computed and checked against hashlib, a global-memory shard) satisfy every constraint and balance every bus."""
import hashlib

import numpy as np
import pytest

import machine_check as MC
import pyoracle as orc
from sp1_amd.machines import public_values as PVM
from sp1_amd.machines import riscv as R
from sp1_amd.machines import riscv_more as M
from sp1_amd.machines import riscv_more_trace as MT
from sp1_amd.machines import riscv_trace as RT

PUBLICS = np.zeros(M.PV_NUM_ELTS, dtype=np.uint64)


@pytest.mark.parametrize("name", sorted(M.MORE_RECORDED))
def test_counts_equal_the_reference_tables(name):
    cols, cons, inter = M.MORE_RECORDED[name]
    s = R.stats(name)
    assert (s["columns"], s["constraints"]) == (cols, cons)
    if inter is not None:
        assert s["interactions"] == inter
    if name in R.RECORDED:                                   # the three chips riscv.py already listed from the reference's tables
        assert R.RECORDED[name] == (cols, cons, inter)


def _check(machine, tabs, publics):
    """Every constraint on every row (asserted) and the exact bus tally of the chips and of the record's eval_public_values."""
    return MC.check_exact(machine, tabs, publics, PVM.program())


CORE = {"Add": 3, "Addi": 5, "Sub": 2, "Bitwise": 3, "Lt": 3, "Mul": 3, "DivRem": 24, "Ecall": 12, "UType": 8, "LoadWord": 3, "LoadByte": 3,
        "StoreWord": 3, "StoreByte": 3, "Branch": 5, "Jal": 2, "Jalr": 2}


@pytest.mark.parametrize("K,seed,clk0", [(3, 5, 1), (2, 6, (1 << 24) - 8 * 40 + 1)])
def test_executed_core_traces_with_divrem_and_ecalls(K, seed, clk0):
    """DIV / DIVU / REM / REMU / DIVW / DIVUW / REMW / REMUW (x0 operands give divisions by zero) and ECALLs — with and without a
    table of their own (SyscallCore rows, Global sends), the clock advancing by 8 + 256 — next to the other instructions."""
    machine, tabs, publics = RT.generate(CORE, K=K, seed=seed, clk0=clk0)
    assert {a.name for a, _ in machine} == set(RT.CORE_CLUSTER)
    assert all(tabs[n][1].shape[0] for n in ("DivRem", "SyscallInstrs", "SyscallCore", "Global", "MemoryLocal"))
    assert not _check(machine, tabs, publics)
    ops = set(tabs["Program"][0][:, 3].tolist())
    assert {R.OPC[o] for o in ("DIV", "DIVU", "REM", "REMU", "DIVW", "DIVUW", "REMW", "REMUW", "ECALL")} <= ops


def test_divrem_special_cases():
    """Overflow (MIN / -1, both widths), division by zero, every sign combination, operands that only differ above bit 31."""
    air, _ = R.chip("DivRem")
    MIN64, MIN32 = -(1 << 63), -(1 << 31)
    cases = []
    for name in RT.ALU_KINDS["DivRem"]:
        for b, c in ((MIN64, -1), (MIN32, -1), (MIN32 & 0xFFFFFFFF, 0xFFFFFFFF), (7, 0), (-7, 0), (0, 0), (-7, 2), (7, -2), (-7, -2), (7, 2),
                     (1 << 40 | 5, 1 << 35 | 3), (-1, 1), (MIN64, 1), (12345678901234, -987654321), (-(1 << 62), 3), ((1 << 63) - 1, -1)):
            cases.append((R.OPC[name], b, c))
    rows = RT.divrem_rows(air.layout, air.main_width, RT.pad32(len(cases)), [c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
    cv = MC.constraint_values(air, None, rows.astype(np.uint64), PUBLICS)
    assert not cv.any(), sorted(set(np.argwhere(cv != 0)[:, 1]))[:8]
    # semantics of the written value against Python's own arithmetic (RISC-V spec: truncating division, x / 0 = -1, x % 0 = x)
    for op, b, c in cases:
        name = RT.OPC_NAME[op]
        a, q, r = RT.divrem_result(op, b, c)
        if name in ("DIVU", "REMU") and c & RT.U64:
            assert (q, r) == divmod(b & RT.U64, c & RT.U64)
        if name in ("DIV", "REM") and c and not (b == MIN64 and c == -1):
            assert RT._S64(q) * c + RT._S64(r) == b and abs(RT._S64(r)) < abs(c)
    # a wrong quotient limb is caught
    bad = rows.copy()
    bad[3, air.layout["quotient"]] = (bad[3, air.layout["quotient"]] + 1) % MC.P
    assert MC.constraint_values(air, None, bad.astype(np.uint64), PUBLICS).any()


def test_keccak_rows_compute_the_permutation():
    st = [0] * 25
    st[0], st[16] = 0x06, 0x80 << 56                        # SHA3-256 padding of the empty message (rate 136 bytes)
    _, out = MT.keccak_f_rows(st)
    assert b"".join(v.to_bytes(8, "little") for v in out[:4]) == hashlib.sha3_256(b"").digest()


def test_precompile_shard_satisfies_every_chip_and_balances():
    machine, tabs, publics = MT.precompile_shard(3, seed=1)
    names = [a.name for a, _ in machine]
    # the reference's Keccak cluster (riscv/mod.rs:L560-L578): the three preprocessed chips, SyscallPrecompile, MemoryLocal, Global + the two
    assert names == ["Byte", "Global", "KeccakPermute", "KeccakPermuteControl", "MemoryLocal", "Program", "Range", "SyscallPrecompile"]
    assert frozenset(names) in RT.chip_clusters()
    assert tabs["KeccakPermute"][1].shape == (96, 2640) and not tabs["Program"][1].any()
    # a precompile shard stands in the program's initial state: timestamp 1, pc = entry, not an execution shard
    assert PVM.get(publics, "initial_timestamp") == PVM.get(publics, "last_timestamp") == [0, 0, 0, 1]
    assert PVM.get(publics, "pc_start") == PVM.get(publics, "next_pc") and PVM.get(publics, "is_execution_shard") == 0
    assert not _check(machine, tabs, publics)
    # one flipped state bit in one round: the constraints (or the Keccak bus) notice
    air = R.chip("KeccakPermute")[0]
    for col in (air.layout["keccak.a_prime.2.3"] + 17, air.layout["keccak.a_prime_prime.1.1"], air.layout["keccak.c.4"] + 63):
        t = {k: (p, m.clone()) for k, (p, m) in tabs.items()}
        t["KeccakPermute"][1][30, col] = (t["KeccakPermute"][1][30, col] + 1) % MC.P
        try:
            imbalance = _check(machine, t, publics)
        except AssertionError:
            continue
        assert imbalance, col


@pytest.mark.parametrize("with_zero", [True, False])
def test_memory_shard_satisfies_every_chip_and_balances(with_zero):
    machine, tabs, publics = MT.memory_shard(40, seed=2, with_zero=with_zero)
    assert [a.name for a, _ in machine] == sorted(RT.MEMORY_CLUSTER)
    assert PVM.get(publics, "global_init_count") == PVM.get(publics, "global_finalize_count") == 40
    assert (PVM.get(publics, "previous_init_addr") == [0, 0, 0]) == with_zero
    assert not _check(machine, tabs, publics)
    # the ends of the two address chains are public values: another last address, count or first address does not balance
    for word in (PVM.PV["last_init_addr"], PVM.PV["global_finalize_count"], PVM.PV["previous_finalize_addr"] + 1):
        pv = publics.clone()
        pv[word] = (pv[word] + 1) % (1 << 16)
        assert _check(machine, tabs, pv), word
    t = {k: (p, m.clone()) for k, (p, m) in tabs.items()}
    lay = R.chip("MemoryGlobalInit")[0].layout
    t["MemoryGlobalInit"][1][5, lay["addr"]], t["MemoryGlobalInit"][1][6, lay["addr"]] = tabs["MemoryGlobalInit"][1][6, lay["addr"]], tabs["MemoryGlobalInit"][1][5, lay["addr"]]
    with pytest.raises(AssertionError):                      # addresses out of order: the comparison constraints fail
        assert not _check(machine, t, publics)


def test_oracle_proves_and_verifies_the_precompile_shard():
    machine, tabs, pv = MT.precompile_shard(2, seed=4)
    chips = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
             for a, i in machine]
    L, lsh, batch, LB, NQ, PW = 17, 12, 8, 1, 5, 4
    prep = orc.JaggedRound([c[3] for c in chips if c[3] is not None], L, lsh, batch, LB)
    ch = orc.Challenger()
    ch.observe(prep.commit)
    v = ch.clone()
    orc.set_gkr_sparse(True)
    try:
        blob = orc.shard_prove(chips, RT.to_monty_np(pv), prep, L, lsh, batch, ch, LB, NQ, PW)
    finally:
        orc.set_gkr_sparse(False)
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
              for a, i in machine]
    pvp = PVM.verifier_program()
    assert orc.shard_verify(shapes, prep.commit, blob, L, lsh, v.clone(), LB, NQ, PW, pv_program=pvp) == 0
    assert orc.shard_verify(shapes, prep.commit, blob, L, lsh, v.clone(), LB, NQ, PW) == 104      # not a zero-sum shard
    bad = bytearray(blob)
    bad[len(bad) // 3] ^= 1
    assert orc.shard_verify(shapes, prep.commit, bytes(bad), L, lsh, v.clone(), LB, NQ, PW, pv_program=pvp) != 0


def test_vectorised_keccak_rounds_equal_the_scalar_ones():
    import torch
    rng = np.random.default_rng(8)
    pre = rng.integers(-(1 << 63), (1 << 63) - 1, size=(3, 25), dtype=np.int64)
    rounds, post = MT.keccak_round_tensors(torch.as_tensor(pre))
    for e in range(3):
        want_rows, want_post = MT.keccak_f_rows([int(v) & MT.U64 for v in pre[e]])
        assert [int(v) & MT.U64 for v in post[e]] == want_post
        for rnd in (0, 7, 23):
            assert [[int(v) & MT.U64 for v in row] for row in rounds[rnd]["a_prime"][e]] == want_rows[rnd]["a_prime"]
            assert int(rounds[rnd]["appp00"][e]) & MT.U64 == want_rows[rnd]["appp00"]
