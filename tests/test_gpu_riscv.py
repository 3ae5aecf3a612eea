"""GPU parity (-m gpu) on the RISC-V core chips (sp1_amd/machines/riscv.py: 30 chips of the rv64im machine transcribed
from the reference's `Air::eval` bodies) over traces that sp1_amd/machines/riscv_trace.py executes: `sp1hip_prove_shard`
bytes == the oracle prover's, and the oracle's full verify_shard (zerocheck closing equation with these constraint
programs + LogUp-GKR interaction check with these interactions) accepts; at 1/64 of the reference's recorded core shard
with production parameters the verifier accepts and a proof over a corrupted trace is rejected."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from sp1_amd.machines import public_values as PVM  # noqa: E402
from sp1_amd.machines import riscv_trace as RT  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))

SMALL = {"Add": 5, "Addi": 7, "Sub": 3, "Bitwise": 6, "Lt": 6, "Mul": 6, "ShiftLeft": 6, "ShiftRight": 8, "Addw": 3, "Subw": 3,
         "UType": 12, "LoadByte": 14, "LoadHalf": 5, "LoadWord": 5, "LoadDouble": 5, "StoreByte": 10, "StoreHalf": 5,
         "StoreWord": 5, "StoreDouble": 5, "Branch": 12, "Jal": 4, "Jalr": 5}


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _shapes_only(machine):
    return [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
            for a, i in machine]


@pytest.mark.parametrize("K,seed,clk0", [(2, 3, 1), (3, 4, (1 << 24) - 8 * 150 + 1)])
def test_riscv_shard_proof_matches_oracle(api, K, seed, clk0):
    """Every chip incl. MemoryBump / StateBump rows (second case: the clock crosses a 2^24 boundary mid-shard)."""
    import core_real
    LB, NQ, PW = 1, 5, 4
    L, lsh, batch = 17, 12, 8                               # the Range table has 2^17 rows
    machine, tabs, publics = RT.generate(SMALL, K=K, seed=seed, clk0=clk0, device="cuda")
    pv = RT.to_monty_np(publics)                            # the shard's own public values: they close its buses (eval_public_values)
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
            for a, i in machine]
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    assert {a.name for a, _ in machine} == set(RT.CORE_CLUSTER)          # the core shape cluster; SyscallCore, DivRem, ... at height zero
    assert all((tabs[n][1].shape[0] > 0) == (clk0 > 1) for n in ("MemoryBump", "StateBump"))
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, LB)
    jp = api.JaggedProver(L, lsh, batch, LB)
    g_commit, g_prep = jp.commit_multilinears([d[3] for d in dev if d[3] is not None])
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
    assert orc.shard_verify(_shapes_only(machine), g_commit, got, L, lsh, v_ch.clone(), LB, NQ, PW, pv_program=PVM.verifier_program()) == 0
    assert orc.shard_verify(_shapes_only(machine), g_commit, got, L, lsh, v_ch.clone(), LB, NQ, PW) == 104   # the cumulative sum is not zero


def test_riscv_shard_bytes_at_1_256_of_the_recorded_shape_match_the_oracle(api):
    """whole-proof byte equality at 1/256 of the recorded core shard (the bench's machine, the real Global chip
    included, production parameters: blowup 4, 124 queries, 16-bit PoW) — the jagged-aware oracle proves it in a few seconds."""
    import core_real
    machine, tabs, publics = core_real.machine_only(scale=1 / 256, seed=9, device="cuda")
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
            for a, i in machine]
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    L, lsh, batch = 17, 14, 32
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, 2)
    commit, prep = api.JaggedProver(L, lsh, batch, 2).commit_multilinears([d[3] for d in dev if d[3] is not None])
    assert np.array_equal(commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(commit)
    g_ch.observe(commit)
    orc.set_gkr_sparse(True)
    try:
        want = orc.shard_prove(host, RT.to_monty_np(publics), o_prep, L, lsh, batch, o_ch, 2, 124, 16)
    finally:
        orc.set_gkr_sparse(False)
    got = api.prove_shard(dev, RT.to_monty_np(publics), prep, L, lsh, batch, g_ch)
    assert got == want and np.array_equal(g_ch.state(), o_ch.state())


def test_riscv_shard_bytes_at_a_sixteenth_of_the_recorded_shape_match_the_oracle(api):
    """Whole-proof byte equality at 1/16 of the recorded core shard (2.4e7 trace cells, max_log_row_count 20: the size of bench.py's
    cpu_baseline sample) — multi-block sums in every kernel, the fused pieces and the bivariate rounds at scale; the jagged-aware
    oracle proves it in a few seconds on the box's 16 threads."""
    import core_real
    machine, tabs, publics = core_real.machine_only(scale=1 / 16, seed=13, device="cuda")
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
            for a, i in machine]
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    del tabs
    L, lsh, batch = 20, 19, 32
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, 2)
    commit, prep = api.JaggedProver(L, lsh, batch, 2).commit_multilinears([d[3] for d in dev if d[3] is not None])
    assert np.array_equal(commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(commit)
    g_ch.observe(commit)
    orc.set_gkr_sparse(True)
    try:
        want = orc.shard_prove(host, RT.to_monty_np(publics), o_prep, L, lsh, batch, o_ch, 2, 124, 16)
    finally:
        orc.set_gkr_sparse(False)
    got = api.prove_shard(dev, RT.to_monty_np(publics), prep, L, lsh, batch, g_ch)
    assert got == want and np.array_equal(g_ch.state(), o_ch.state())


def test_riscv_shard_bytes_at_a_quarter_of_the_recorded_shape_match_the_oracle(api):
    """VERDICT r4 #7b: whole-proof byte equality at 1/4 of the recorded core shard (9.3e7 trace cells, max_log_row_count 21, every one
    of its 33 chips incl. DivRem / SyscallInstrs / SyscallCore) with production parameters — the largest size the oracle proves in
    tens of seconds on the box's 16 threads."""
    import core_real
    machine, tabs, publics = core_real.machine_only(scale=1 / 4, seed=23, device="cuda")
    assert all(tabs[n][1].shape[0] for n in ("DivRem", "SyscallInstrs", "SyscallCore", "Global"))
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
            for a, i in machine]
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    del tabs
    L, lsh, batch = 21, 20, 32
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, 2)
    commit, prep = api.JaggedProver(L, lsh, batch, 2).commit_multilinears([d[3] for d in dev if d[3] is not None])
    assert np.array_equal(commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(commit)
    g_ch.observe(commit)
    orc.set_gkr_sparse(True)
    try:
        want = orc.shard_prove(host, RT.to_monty_np(publics), o_prep, L, lsh, batch, o_ch, 2, 124, 16)
    finally:
        orc.set_gkr_sparse(False)
    got = api.prove_shard(dev, RT.to_monty_np(publics), prep, L, lsh, batch, g_ch)
    assert got == want and np.array_equal(g_ch.state(), o_ch.state())


def test_riscv_shard_at_a_sixty_fourth_of_the_recorded_shape_verifies(api):
    """Production parameters (blowup 4, 124 queries, 16-bit PoW); 1/64 of the recorded heights, the real Global chip
    included (the bench's shard). One wrong cell in the Bitwise table -> the verifier rejects; so does one wrong
    coordinate of a Global row's curve point."""
    import core_real
    chips, meta = core_real.build_real_shard(scale=1 / 64, seed=5)
    assert len(meta["real_chips"]) == 34 and not meta["synthetic_chips"] and "Global" not in meta["empty_chips"]
    PUBLICS, pvp = meta["publics"], PVM.verifier_program()
    L, lsh = 17, 16
    jp = api.JaggedProver(L, lsh, 32, 2)
    commit, prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
    ch, v = api.DuplexChallenger(), orc.Challenger()
    ch.observe(commit)
    v.observe(commit)
    proof = api.prove_shard(chips, PUBLICS, prep, L, lsh, 32, ch)
    shapes = _shapes_only([(a, i) for a, i, _, _ in chips])
    assert orc.shard_verify(shapes, commit, proof, L, lsh, v, 2, 124, 16, pv_program=pvp) == 0
    assert np.array_equal(v.state(), ch.state())
    k = [a.name for a, _, _, _ in chips].index("Bitwise")
    a, i, m, p = chips[k]
    bad = m.words.clone()
    col = a.layout["result"]
    bad[col * m.height + 7] ^= 1 << 20
    chips2 = list(chips)
    chips2[k] = (a, i, api.ColMajor(bad, m.height, m.width), p)
    ch, v = api.DuplexChallenger(), orc.Challenger()
    ch.observe(commit)
    v.observe(commit)
    proof = api.prove_shard(chips2, PUBLICS, prep, L, lsh, 32, ch)
    assert orc.shard_verify(shapes, commit, proof, L, lsh, v, 2, 124, 16, pv_program=pvp) != 0
    k = [a.name for a, _, _, _ in chips].index("Global")
    a, i, m, p = chips[k]
    bad = m.words.clone()
    bad[(a.layout["interaction.y_coordinate"] + 3) * m.height + 11] ^= 1 << 9
    chips3 = list(chips)
    chips3[k] = (a, i, api.ColMajor(bad, m.height, m.width), p)
    ch, v = api.DuplexChallenger(), orc.Challenger()
    ch.observe(commit)
    v.observe(commit)
    proof = api.prove_shard(chips3, PUBLICS, prep, L, lsh, 32, ch)
    assert orc.shard_verify(shapes, commit, proof, L, lsh, v, 2, 124, 16, pv_program=pvp) != 0


def _small_riscv_case(api, seed=21):
    import core_real
    L, lsh, batch = 17, 12, 8
    machine, tabs, publics = RT.generate(SMALL, K=2, seed=seed, device="cuda")
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
            for a, i in machine]
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    return machine, host, dev, L, lsh, batch, RT.to_monty_np(publics)


@pytest.mark.parametrize("env", [{"SP1HIP_ZC_BIVARIATE": "0"}, {"SP1HIP_ZC_FORK": "0"}, {"SP1HIP_ZC_BIVARIATE": "0", "SP1HIP_ZC_FORK": "0"},
                                 {"SP1HIP_ZC_MACRO": "0"}, {"SP1HIP_WAIT": "spin"}])
def test_riscv_shard_with_hinted_chip_under_every_zerocheck_switch(api, monkeypatch, env):
    """the sequential rounds (zc_macro_kernel<true, KIND>: the round-0 form of the fused Poseidon2 / septic pieces) and the
    single-stream launch path, on a machine with a HINTed chip (Global); also hints ignored (the interpreter evaluates the
    sub-AIRs) — same bytes as the oracle in every configuration."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    LB, NQ, PW = 1, 5, 4
    machine, host, dev, L, lsh, batch, pv = _small_riscv_case(api)
    assert any(a.name == "Global" for a, _ in machine)
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, LB)
    g_commit, g_prep = api.JaggedProver(L, lsh, batch, LB).commit_multilinears([d[3] for d in dev if d[3] is not None])
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(o_prep.commit)
    g_ch.observe(g_commit)
    orc.set_gkr_sparse(True)
    try:
        want = orc.shard_prove(host, pv, o_prep, L, lsh, batch, o_ch, LB, NQ, PW)
    finally:
        orc.set_gkr_sparse(False)
    got = api.prove_shard(dev, pv, g_prep, L, lsh, batch, g_ch, LB, NQ, PW)
    assert got == want and np.array_equal(g_ch.state(), o_ch.state())


def test_riscv_pool_proofs_match_the_oracle(api):
    """several provers in flight take the non-forked launch path; pool proofs of the rv64im machine (hinted Global chip)
    are checked byte for byte against the ORACLE, three in flight, repeated (the second round of proofs also runs on the learned
    hand-over timeline: common.hpp WaitPlan)."""
    LB, NQ, PW = 1, 5, 4
    machine, host, dev, L, lsh, batch, pv = _small_riscv_case(api, seed=22)
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, LB)
    pk = api.ProvingKey([d[3] for d in dev if d[3] is not None], L, lsh, batch, log_blowup=LB, num_queries=NQ, pow_bits=PW)
    direct = pk.prove_shard(dev, pv)
    pool = api.ProverPool(3)
    tickets = [pool.submit(pk, dev, pv) for _ in range(9)]
    for t in tickets:
        proof, _ = pool.wait(t)
        assert proof == direct
    pool.close()
    # the oracle prover from the same transcript head (vk.observe_into = commit, pc_start, septic x / y, flag, 6 zeros)
    assert np.array_equal(pk.preprocessed_commit, o_prep.commit)
    o_ch = orc.Challenger()
    o_ch.observe(np.concatenate([o_prep.commit, np.zeros(3 + 14 + 7, np.uint32)]))
    g_head = api.DuplexChallenger()
    pk.observe_into(g_head)
    assert np.array_equal(o_ch.state(), g_head.state())
    orc.set_gkr_sparse(True)
    try:
        want = orc.shard_prove(host, pv, o_prep, L, lsh, batch, o_ch, LB, NQ, PW)
    finally:
        orc.set_gkr_sparse(False)
    assert direct == want
