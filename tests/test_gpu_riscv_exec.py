"""GPU parity (-m gpu) on REAL guest programs: the reference's fibonacci / keccak guests (bench/programs/*.elf.gz) executed by the
rv64im executor of libsp1hip.so, every shard of the run — core shards, the KECCAK_PERMUTE precompile shard, the memory shard —
proved by `sp1hip_prove_shard` with the shard's own public values: bytes == the oracle prover's on the same tables, and the
oracle's verify_shard accepts. One larger fibonacci shard (2^18 cycles) with production parameters."""
import os
import struct
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from sp1_amd.machines import public_values as PVM  # noqa: E402
from sp1_amd.machines import riscv_exec as X  # noqa: E402
from sp1_amd.machines import riscv_trace as RT  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bench"))


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _elf(name):
    return X.guest_file(name + ".elf")


def _shapes_only(machine):
    return [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
            for a, i in machine]


def prove_both(api, machine, tabs, publics, L, lsh, batch, LB, NQ, PW, verify=True):
    import core_real
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
            for a, i in machine]
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    pv = RT.to_monty_np(publics)                          # public values travel as Montgomery words, like the tables
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, LB)
    g_commit, g_prep = api.JaggedProver(L, lsh, batch, LB).commit_multilinears([d[3] for d in dev if d[3] is not None])
    assert np.array_equal(g_commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(o_prep.commit)
    g_ch.observe(g_commit)
    v_ch = o_ch.clone()
    orc.set_gkr_sparse(True)                                # the jagged-aware oracle prover (bytes equal to the dense one)
    try:
        want = orc.shard_prove(host, pv, o_prep, L, lsh, batch, o_ch, LB, NQ, PW)
    finally:
        orc.set_gkr_sparse(False)
    got = api.prove_shard(dev, pv, g_prep, L, lsh, batch, g_ch, LB, NQ, PW)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    if verify:      # the reference's statement: the chip set is a shape cluster of the machine, the cumulative sum is the public values'
        assert frozenset(a.name for a, _ in machine) in RT.chip_clusters()
        assert orc.shard_verify(_shapes_only(machine), g_commit, got, L, lsh, v_ch, LB, NQ, PW, pv_program=PVM.verifier_program()) == 0


@pytest.mark.parametrize("program,stdin,max_cycles,kinds", [
    ("fibonacci", [struct.pack("<Q", 300)], 3000, ["core", "core", "core", "memory"]),
    ("keccak", [bytes(300)], 6000, ["core", "core", "keccak", "memory"]),
    ("sha2", [bytes(100)], 1 << 20, ["core", "sha_extend", "sha_compress", "memory"]),
    ("poseidon2", [struct.pack("<Q", 20)], 1 << 20, ["core", "poseidon2", "sha_extend", "sha_compress", "memory"]),
])
def test_every_shard_of_a_real_program_matches_the_oracle(api, program, stdin, max_cycles, kinds, monkeypatch):
    monkeypatch.setenv("SP1HIP_ZC_MUL_MIN_ROWS", "0")                    # the fused MulOperation piece on these small tables too
    ex = X.Executor(_elf(program), stdin=stdin)
    seen, gevs = [], []
    for kind, machine, tabs, publics, gev, sh in X.program_shards(ex, max_cycles, device="cuda"):
        prove_both(api, machine, tabs, publics, 17, 12, 8, 1, 5, 4)     # the Range table has 2^17 rows
        seen.append(kind)
        gevs.append(gev)
    assert seen == kinds
    assert not X.global_events_balance(gevs + [X.image_events(ex)])


def test_the_big_integer_precompile_shards_match_the_oracle(api):
    """UINT256_MUL, SECP256K1_DOUBLE and SECP256K1_ADD from a hand-assembled program (2G, then G + 2G, then a 256-bit modular
    product): the FieldOpCols chips — 23k-instruction constraint programs, 1.1k byte / range lookups per row — through the GPU prover."""
    import rv_asm as A
    M64 = (1 << 64) - 1
    G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
    words = lambda v: b"".join(struct.pack("<Q", (v >> (64 * i)) & M64) for i in range(4))
    data = words(G[0]) + words(G[1]) + words(G[0]) + words(G[1]) + words(0x1234567890ABCDEF << 130 | 77) + words((1 << 255) + 12345) + words((1 << 256) - 189)
    prog = A.li(28, 0x78100000)
    prog += [A.enc("addi", 10, 28, 0), A.enc("addi", 11, 0, 0)] + A.li(5, 0x0000010B) + [A.enc("ecall")]
    prog += [A.enc("addi", 10, 28, 64), A.enc("addi", 11, 28, 0)] + A.li(5, 0x0001010A) + [A.enc("ecall")]
    prog += [A.enc("addi", 10, 28, 128), A.enc("addi", 11, 28, 160)] + A.li(5, 0x0001011D) + [A.enc("ecall")]
    ex = X.Executor(A.elf(prog + A.halt(0), data=data + bytes(32)), stdin=[])
    seen, gevs = [], []
    for kind, machine, tabs, publics, gev, sh in X.program_shards(ex, 1 << 20, device="cuda"):
        prove_both(api, machine, tabs, publics, 17, 12, 8, 1, 5, 4)
        seen.append(kind)
        gevs.append(gev)
    assert seen == ["core", "uint256", "secp256k1_add", "secp256k1_double", "memory"]
    assert not X.global_events_balance(gevs + [X.image_events(ex)])


@pytest.mark.parametrize("mul_min_rows", ["0", None])
def test_a_fibonacci_shard_with_production_parameters_matches_the_oracle(api, mul_min_rows, monkeypatch):
    """2^18 cycles of the reference's fibonacci guest (1.3e7 trace cells), blowup 4, 124 queries, 16-bit PoW — with the fused
    MulOperation piece forced on (it is honoured from 2^16 Mul rows; this shard has 58k) and with the default rule."""
    if mul_min_rows is not None:
        monkeypatch.setenv("SP1HIP_ZC_MUL_MIN_ROWS", mul_min_rows)
    ex = X.Executor(_elf("fibonacci"), stdin=[struct.pack("<Q", 40000)])
    sh = ex.run_shard(1 << 18)
    assert sh.cycles == 1 << 18 and not sh.halted
    machine, tabs, publics = X.shard_tables(ex, sh, device="cuda")
    prove_both(api, machine, tabs, publics, 18, 17, 32, 2, 124, 16)


def test_a_corrupted_real_trace_is_rejected(api):
    """The GPU prover proves what it is given; the verifier is what refuses a shard whose Mul row is wrong."""
    import core_real
    ex = X.Executor(_elf("fibonacci"), stdin=[struct.pack("<Q", 100)])
    sh = ex.run_shard(1 << 20)
    machine, tabs, publics = X.shard_tables(ex, sh, device="cuda")
    from sp1_amd.machines import riscv as R
    col = R.chip("Mul")[0].layout["a"]
    tabs["Mul"][1][3, col] = (tabs["Mul"][1][3, col] + 1) % RT.P
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    pv = RT.to_monty_np(publics)
    L, lsh, batch = 17, 12, 8
    commit, prep = api.JaggedProver(L, lsh, batch, 1).commit_multilinears([d[3] for d in dev if d[3] is not None])
    ch = api.DuplexChallenger()
    ch.observe(commit)
    proof = api.prove_shard(dev, pv, prep, L, lsh, batch, ch, 1, 5, 4)
    v_ch = orc.Challenger()
    v_ch.observe(commit)
    assert orc.shard_verify(_shapes_only(machine), commit, proof, L, lsh, v_ch, 1, 5, 4, pv_program=PVM.verifier_program()) != 0


def test_a_proof_for_other_public_values_is_rejected(api):
    """The GPU prover proves with the public values it is given (the reference's prover samples `pv_challenge` and drops it,
    logup_gkr/prover.rs:L93); the VERIFIER derives the cumulative sum the LogUp-GKR output must have from them: a shard proved
    for a different entry pc, digest or clock fails with the cumulative-sum code, a public word that breaks eval_public_values'
    own constraints with the public-values code."""
    import core_real
    ex = X.Executor(_elf("fibonacci"), stdin=[struct.pack("<Q", 100)])
    sh = ex.run_shard(1 << 20)
    machine, tabs, publics = X.shard_tables(ex, sh, device="cuda")
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    L, lsh, batch = 17, 12, 8
    commit, prep = api.JaggedProver(L, lsh, batch, 1).commit_multilinears([d[3] for d in dev if d[3] is not None])
    for word, code in ((None, 0), (PVM.PV["pc_start"], 104), (PVM.PV["global_cumulative_sum"] + 2, 104), (PVM.PV["global_count"], 104),
                       (PVM.PV["is_execution_shard"], 109), (PVM.PV["last_timestamp"] + 3, 109), (PVM.NUM_PV_ELTS + 3, 4)):
        pv = publics.clone()
        if word is not None:
            pv[word] = (pv[word] + 1) % RT.P
        ch = api.DuplexChallenger()
        ch.observe(commit)
        proof = api.prove_shard(dev, RT.to_monty_np(pv), prep, L, lsh, batch, ch, 1, 5, 4)
        v_ch = orc.Challenger()
        v_ch.observe(commit)
        assert orc.shard_verify(_shapes_only(machine), commit, proof, L, lsh, v_ch, 1, 5, 4, pv_program=PVM.verifier_program()) == code, word
