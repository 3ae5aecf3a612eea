"""GPU parity (-m gpu) on the chips of riscv_more.py: whole-proof bytes of `sp1hip_prove_shard` == the oracle prover's on
 (1) an executed core shard with DIV/REM and ECALL instructions (DivRem, SyscallInstrs, SyscallCore next to the other chips),
 (2) a Keccak precompile shard — the 2,640-column KeccakPermute chip, its controller, SyscallPrecompile, MemoryLocal, Global —
     the wide-chip regime of the zerocheck,
 (3) a global-memory shard (MemoryGlobalInit / Finalize),
and the oracle's full verify_shard accepts them."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from sp1_amd.machines import public_values as PVM  # noqa: E402
from sp1_amd.machines import riscv_more_trace as MT  # noqa: E402
from sp1_amd.machines import riscv_trace as RT  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _prove_both(api, machine, tabs, publics, L=17, lsh=12, batch=8, LB=1, NQ=5, PW=4, check_verifier=True):
    import core_real
    PUBLICS = RT.to_monty_np(publics.cpu())                 # the shard's own public values (they close its buses)
    host = [(a, i, RT.to_monty_np(tabs[a.name][1].cpu()), RT.to_monty_np(tabs[a.name][0].cpu()) if tabs[a.name][0] is not None else None)
            for a, i in machine]
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1].cuda()), core_real.to_col_major(tabs[a.name][0].cuda()) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, LB)
    g_commit, g_prep = api.JaggedProver(L, lsh, batch, LB).commit_multilinears([d[3] for d in dev if d[3] is not None])
    assert np.array_equal(g_commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(o_prep.commit)
    g_ch.observe(g_commit)
    v_ch = o_ch.clone()
    orc.set_gkr_sparse(True)
    try:
        want = orc.shard_prove(host, PUBLICS, o_prep, L, lsh, batch, o_ch, LB, NQ, PW)
    finally:
        orc.set_gkr_sparse(False)
    got = api.prove_shard(dev, PUBLICS, g_prep, L, lsh, batch, g_ch, LB, NQ, PW)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    if check_verifier:
        shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
                  for a, i in machine]
        assert orc.shard_verify(shapes, g_commit, got, L, lsh, v_ch, LB, NQ, PW, pv_program=PVM.verifier_program()) == 0
    return got


def test_core_shard_with_divrem_and_ecalls_matches_oracle(api):
    counts = {"Add": 3, "Addi": 5, "Sub": 2, "Bitwise": 3, "Lt": 3, "Mul": 3, "DivRem": 24, "Ecall": 12, "UType": 8, "LoadWord": 3, "LoadByte": 3,
              "StoreWord": 3, "StoreByte": 3, "Branch": 5, "Jal": 2, "Jalr": 2}
    machine, tabs, publics = RT.generate(counts, K=3, seed=5, device="cuda")
    assert all(tabs[n][1].shape[0] for n in ("DivRem", "SyscallInstrs", "SyscallCore"))
    _prove_both(api, machine, tabs, publics)


@pytest.mark.parametrize("n_events,env", [(2, {}), (12, {}), (12, {"SP1HIP_ZC_BIVARIATE": "0"}), (12, {"SP1HIP_ZC_MACRO": "0"}),
                                          (12, {"SP1HIP_ZC_KECCAK3": "1"}), (12, {"SP1HIP_ZC_KECCAK3": "1", "SP1HIP_ZC_BIVARIATE": "0"})])
def test_keccak_precompile_shard_matches_oracle(api, monkeypatch, n_events, env):
    """The wide chip through every zerocheck path: 48 / 288 rows of 2,640 columns (288 rows = the multi-block forms of the round
    kernels), sequential rounds, hints ignored."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    machine, tabs, publics = MT.precompile_shard(n_events, seed=3, device="cuda")
    assert tabs["KeccakPermute"][1].shape[1] == 2640
    _prove_both(api, machine, tabs, publics, check_verifier=(n_events == 2))


def test_keccak_precompile_shard_at_production_parameters(api):
    """1/64 of the bench's precompile shard (80 permutations, 1,920 rows of the wide chip), blowup 4, 124 queries, 16-bit PoW."""
    machine, tabs, publics = MT.precompile_shard(80, seed=7, device="cuda")
    _prove_both(api, machine, tabs, publics, L=17, lsh=15, batch=32, LB=2, NQ=124, PW=16, check_verifier=False)


def test_memory_shard_matches_oracle(api):
    machine, tabs, publics = MT.memory_shard(300, seed=2, device="cuda")
    _prove_both(api, machine, tabs, publics)
