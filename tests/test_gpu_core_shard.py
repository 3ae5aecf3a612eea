"""GPU parity (-m gpu) on the bench workload: the core-SHAPED synthetic shard of bench/core_shard.py (33 chips with the
widths / constraint counts / interaction counts of a recorded RISC-V core shard). At 1/4096 of the CORE area the traces
are checked row by row (every constraint zero, lookups balanced), the proof bytes equal the oracle prover's and the
pinned verifier accepts; at 1/64 of the area, with the production parameters, the succinct verifier accepts."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bench"))

import machine_check as MC  # noqa: E402
import pyoracle as orc  # noqa: E402

CORE_AREA = (1 << 28) + (1 << 27)


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _shapes_only(chips):
    return [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
            for a, i, _, _ in chips]


def test_core_shaped_shard_is_satisfying_and_proof_matches_oracle(api):
    from core_shard import build_core_shard
    k = 6
    L, lsh = 22 - k, 21 - k
    chips, meta = build_core_shard(CORE_AREA >> (2 * k), L)
    assert meta["chips"] == 33 and meta["interactions"] == 730 and meta["constraints"] == 1605
    host = [(a, i, m.to_row_major_host(), p.to_row_major_host() if p is not None else None) for a, i, m, p in chips]
    bus = []
    for a, i, m, p in host:
        mc, pc = MC.from_monty(m), MC.from_monty(p) if p is not None else np.zeros((m.shape[0], 0), np.uint64)
        assert not MC.constraint_values(a, pc, mc, []).any(), a.name
        bus.append((i, pc, mc))
    assert MC.bus_imbalance(bus) == {}
    LB, NQ, PW = 1, 5, 4
    o_prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, 32, LB)
    jp = api.JaggedProver(L, lsh, 32, LB)
    g_commit, g_prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
    assert np.array_equal(g_commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(o_prep.commit)
    g_ch.observe(g_commit)
    v_ch = o_ch.clone()
    want = orc.shard_prove(host, np.zeros(0, np.uint32), o_prep, L, lsh, 32, o_ch, LB, NQ, PW)
    got = api.prove_shard(chips, [], g_prep, L, lsh, 32, g_ch, LB, NQ, PW)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    assert orc.shard_verify(_shapes_only(chips), g_commit, got, L, lsh, v_ch, LB, NQ, PW) == 0


def test_core_shaped_shard_at_a_64th_of_core_size_verifies(api):
    from core_shard import build_core_shard
    k = 3
    L, lsh = 22 - k, 21 - k
    chips, meta = build_core_shard(CORE_AREA >> (2 * k), L)
    jp = api.JaggedProver(L, lsh, 32, 2)
    commit, prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
    ch, v = api.DuplexChallenger(), orc.Challenger()
    ch.observe(commit)
    v.observe(commit)
    proof = api.prove_shard(chips, [], prep, L, lsh, 32, ch)
    assert orc.shard_verify(_shapes_only(chips), commit, proof, L, lsh, v, 2, 124, 16) == 0
    assert np.array_equal(v.state(), ch.state())


def test_core_shaped_shard_at_core_size_verifies(api):
    """The bench workload at the bench size: one core-shaped shard of 2^28 + 2^27 cells proven by
    sp1hip_prove_shard with the production parameters; the pinned verifier accepts, ends in the prover's transcript state,
    and rejects the proof with one bit flipped. (bench.py runs the same check on its last timed proof.)"""
    from core_shard import build_core_shard
    L, lsh = 22, 21
    chips, meta = build_core_shard(CORE_AREA, L)
    assert meta["area_cells"] > 4.0e8
    jp = api.JaggedProver(L, lsh, 32, 2)
    commit, prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
    ch, v = api.DuplexChallenger(), orc.Challenger()
    ch.observe(commit)
    v.observe(commit)
    proof = api.prove_shard(chips, [], prep, L, lsh, 32, ch)
    shapes = _shapes_only(chips)
    del chips, prep
    assert orc.shard_verify(shapes, commit, proof, L, lsh, v, 2, 124, 16) == 0
    assert np.array_equal(v.state(), ch.state())
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 4
    v2 = orc.Challenger()
    v2.observe(commit)
    assert orc.shard_verify(shapes, commit, bytes(bad), L, lsh, v2, 2, 124, 16) != 0
