"""The hand-transcribed RISC-V core chips (sp1_amd/machines/riscv.py) and their executed traces (riscv_trace.py), on the CPU.

No reference-made RISC-V proof exists in the tree (unlike the recursion machine, tests/test_recursion_machine.py), so the
transcription is pinned by (1) three independent data points per chip that the reference tree DOES hold — column counts
(rv64im_costs.json), `assert_zero` counts (rv64im_complexity.json) and interaction counts (the recorded core shard of
sp1-gpu/crates/logup_gkr/layer_workloads.json, decoded by chip name) — and (2) semantics: traces filled from an rv64im
execution must make every constraint vanish on every row and every bus balance; (3) the oracle's prover and full verifier
run the machine end to end."""
import numpy as np
import pytest

import machine_check as MC
import pyoracle as orc
from sp1_amd.machines import public_values as PVM, riscv as R, riscv_trace as RT

torch = pytest.importorskip("torch")

# the recorded shard predates three range checks the current source has (control_flow/jal/air.rs `next_pc[0] / 4 < 2^14`,
# the same in jalr/air.rs, memory/bump.rs `addr < 32`) and the byte decomposition of y6 in global_interaction.rs: those four
# chips have their CURRENT column and constraint counts, and more interactions than the recording
NEWER_THAN_RECORDING = {"Jal": 18, "Jalr": 22, "MemoryBump": 8, "Global": 12}
FULL = {"Add": 5, "Addi": 7, "Sub": 3, "Bitwise": 6, "Lt": 6, "Mul": 6, "ShiftLeft": 6, "ShiftRight": 8, "Addw": 3, "Subw": 3,
        "UType": 12, "LoadByte": 14, "LoadHalf": 5, "LoadWord": 5, "LoadDouble": 5, "LoadX0": 9, "StoreByte": 10, "StoreHalf": 5,
        "StoreWord": 5, "StoreDouble": 5, "Branch": 12, "Jal": 4, "Jalr": 9}


@pytest.mark.parametrize("name", sorted(R.CHIPS))
def test_chip_counts_match_the_reference_tables(name):
    s = R.stats(name)
    cols, cons, inter = R.RECORDED[name]
    assert (s["columns"], s["constraints"]) == (cols, cons)
    assert s["interactions"] == NEWER_THAN_RECORDING.get(name, inter)


def test_recorded_shard_is_decoded_consistently():
    """Chips in name order; area of the decoded shard below the shard limit 2^28 + 2^27 and within 10 % of it."""
    area = sum(R.RECORDED_ROWS[n] * R.RECORDED[n][0] for n in R.RECORDED_ROWS)
    assert 0.9 * ((1 << 28) + (1 << 27)) < area <= (1 << 28) + (1 << 27)
    assert R.RECORDED_ROWS["Byte"] == 1 << 16 and R.RECORDED_ROWS["Range"] == 1 << 17
    assert sorted(R.RECORDED_ROWS) == sorted(R.RECORDED)


def _check(machine, tabs, publics, with_public_values=True):
    """Every chip's constraints on every row; the buses of the chips AND of the record's eval_public_values (the exact tally)."""
    return MC.check_exact(machine, tabs, publics, PVM.program() if with_public_values else None)


@pytest.mark.parametrize("K,seed,clk0,pc_base", [
    (2, 10, 1, 0x200000),
    (3, 11, (1 << 24) - 8 * 150 + 1, 0x20FF00),          # the clock crosses 2^24 and the pc a 16-bit limb: StateBump / MemoryBump rows
    (1, 12, 9, 0x3FFFE0),
])
def test_executed_traces_satisfy_every_chip_and_balance_every_bus(K, seed, clk0, pc_base):
    machine, tabs, publics = RT.generate(FULL, K=K, seed=seed, clk0=clk0, pc_base=pc_base)
    names = {a.name for a, _ in machine}
    assert names == set(RT.CORE_CLUSTER) and frozenset(names) in RT.chip_clusters()      # the shard IS a shape cluster of the machine
    rows = {n: tabs[n][1].shape[0] for n in names}
    assert all(rows[n] for n in set(FULL) | {"MemoryLocal", "Program", "Byte", "Range", "Global"})
    assert not any(rows[n] for n in ("DivRem", "SyscallCore", "SyscallInstrs", "AluX0"))  # in the cluster, without events: height zero
    assert bool(rows["MemoryBump"]) == (clk0 > (1 << 23)) and bool(rows["StateBump"]) == (clk0 > (1 << 23) or pc_base == 0x3FFFE0)
    assert publics.shape[0] == PVM.PROOF_MAX_NUM_PVS and not publics[PVM.NUM_PV_ELTS:].any()
    assert not _check(machine, tabs, publics)
    # the chips alone do not balance: the initial / final CPU state, the ends of the accumulation chain and the range checks of
    # the public limbs are the record's own messages (eval_public_values)
    open_ = _check(machine, tabs, publics, with_public_values=False)
    assert {k[0] for k in open_} == {R.STATE, R.GLOBAL_ACC, R.BYTE}


def test_the_shape_clusters_are_the_reference_list():
    """riscv/mod.rs:L560-L803 without `mprotect`: 1 + 6 + 1 core clusters, the special one, the memory cluster, 22 precompile
    clusters; every cluster holds the three preprocessed chips and Global; smallest_cluster picks like MachineShape's."""
    clusters = RT.chip_clusters()
    assert len(clusters) == 8 + 1 + 1 + 22 and len(set(clusters)) == len(clusters)
    assert all({"Program", "Byte", "Range", "Global"} <= c for c in clusters)
    assert len(RT.CORE_CLUSTER) == 34 and len(RT.MEMORY_CLUSTER) == 6
    assert RT.smallest_cluster({"Add", "Byte"}) == frozenset(RT.CORE_CLUSTER)
    assert RT.smallest_cluster({"KeccakPermute"}) == frozenset(RT.PRECOMPILE_CLUSTERS[8])
    assert RT.smallest_cluster({"Add", "Poseidon2"}) == frozenset(RT.CORE_CLUSTER + ["Poseidon2"])
    assert RT.smallest_cluster({"Add", "Poseidon2", "Uint256Ops"}) == max(clusters, key=len)
    for c in clusters:                                        # every name is a transcribed chip
        for n in c:
            R.chip(n)


def test_a_wrong_cell_is_caught():
    machine, tabs, publics = RT.generate({"Add": 4, "Bitwise": 4, "LoadByte": 4, "StoreByte": 4, "UType": 8}, K=2, seed=3)
    # a carry bit of Add, a curve coordinate of Global, a byte multiplicity
    for name, row, col in (("Add", 1, R.chip("Add")[0].layout["value"]), ("Global", 2, R.chip("Global")[0].layout["interaction.y_coordinate"] + 5),
                           ("Byte", 0, 3)):
        t = {k: (p, m.clone()) for k, (p, m) in tabs.items()}
        t[name][1][row, col] = (t[name][1][row, col] + 1) % MC.P
        try:
            imbalance = _check(machine, t, publics)
        except AssertionError:
            continue
        assert imbalance, name
    # ... and so is a wrong public value: the entry pc (State bus), the digest (GlobalAccumulation), a timestamp limb (constraints)
    for word in (PVM.PV["pc_start"], PVM.PV["global_cumulative_sum"] + 9, PVM.PV["global_count"], PVM.PV["last_timestamp"] + 2):
        pv = publics.clone()
        pv[word] = (pv[word] + 1) % MC.P
        try:
            imbalance = _check(machine, tabs, pv)
        except AssertionError:
            continue
        assert imbalance, word


def test_oracle_proves_and_verifies_the_riscv_machine():
    machine, tabs, publics = RT.generate(FULL, K=2, seed=3)
    chips = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
             for a, i in machine]
    L, lsh, batch, LB, NQ, PW = 17, 12, 8, 1, 5, 4
    prep = orc.JaggedRound([c[3] for c in chips if c[3] is not None], L, lsh, batch, LB)
    ch = orc.Challenger()
    ch.observe(prep.commit)
    v = ch.clone()
    orc.set_gkr_sparse(True)
    try:
        blob = orc.shard_prove(chips, RT.to_monty_np(publics), prep, L, lsh, batch, ch, LB, NQ, PW)
        other = publics.clone()
        other[PVM.PV["next_pc"]] += 4
        lie = orc.shard_prove(chips, RT.to_monty_np(other), prep, L, lsh, batch, v.clone(), LB, NQ, PW)
    finally:
        orc.set_gkr_sparse(False)
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
              for a, i in machine]
    pvp = PVM.verifier_program()
    assert orc.shard_verify(shapes, prep.commit, blob, L, lsh, v.clone(), LB, NQ, PW, pv_program=pvp) == 0
    # the cumulative sum of the circuit output is MINUS what the public values send: a verifier that expects zero (a machine
    # without eval_public_values) refuses the same bytes, and the real one refuses a proof made for other public values
    assert orc.shard_verify(shapes, prep.commit, blob, L, lsh, v.clone(), LB, NQ, PW) == 104
    assert orc.shard_verify(shapes, prep.commit, lie, L, lsh, v.clone(), LB, NQ, PW, pv_program=pvp) == 104
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 1
    assert orc.shard_verify(shapes, prep.commit, bytes(bad), L, lsh, v.clone(), LB, NQ, PW, pv_program=pvp) != 0
