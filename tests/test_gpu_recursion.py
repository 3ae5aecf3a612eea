"""GPU parity (-m gpu) on the REAL machine: `sp1hip_prove_shard` proves satisfying traces of the reference's recursion
compress machine (sp1_amd/machines/recursion.py — the transcription that the reference's own ShardProof pins in
tests/test_recursion_machine.py): bytes equal the oracle prover's, and the oracle's full verify_shard (zerocheck closing
equation + LogUp-GKR interaction check with the real chips) accepts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from sp1_amd.machines import recursion as R, recursion_trace as RT  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _shapes_only(machine):
    return [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
            for a, i in machine]


def _device_chips(api, machine, tabs):
    return [(a, i, api.ColMajor.from_row_major_host(tabs[a.name][1]), api.ColMajor.from_row_major_host(tabs[a.name][0]))
            for a, i in machine]


@pytest.mark.parametrize("counts,seed,L,lsh,batch", [
    ({"BaseAlu": 70, "ExtAlu": 90, "MemoryConst": 50, "MemoryVar": 40, "Poseidon2WideDeg3": 20, "PrefixSumChecks": 33, "Select": 100},
     1, 8, 7, 4),
    ({"BaseAlu": 600, "ExtAlu": 800, "MemoryConst": 500, "MemoryVar": 650, "Poseidon2WideDeg3": 150, "PrefixSumChecks": 220,
      "Select": 1100}, 2, 11, 9, 32),
])
def test_recursion_shard_proof_matches_oracle(api, counts, seed, L, lsh, batch):
    LB, NQ, PW = 1, 5, 4
    tabs, pv = RT.generate(counts, seed=seed)
    m = R.compress_machine()
    chips = [(a, i, tabs[a.name][1], tabs[a.name][0]) for a, i in m]
    o_prep = orc.JaggedRound([c[3] for c in chips], L, lsh, batch, LB)
    jp = api.JaggedProver(L, lsh, batch, LB)
    dev = _device_chips(api, m, tabs)
    g_commit, g_prep = jp.commit_multilinears([d[3] for d in dev])
    assert np.array_equal(g_commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(o_prep.commit)
    g_ch.observe(g_commit)
    v_ch = o_ch.clone()
    want = orc.shard_prove(chips, pv, o_prep, L, lsh, batch, o_ch, LB, NQ, PW)
    got = api.prove_shard(dev, pv, g_prep, L, lsh, batch, g_ch, LB, NQ, PW)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    assert orc.shard_verify(_shapes_only(m), g_commit, got, L, lsh, v_ch, LB, NQ, PW) == 0


def test_recursion_shard_at_a_sixteenth_of_the_reference_shape_verifies(api):
    """Production parameters (blowup 4, 124 queries, 16-bit PoW, stacking height as in the recursion prover scaled with
    the shape): 1/16 of the table heights of the reference's real compress proof (5.6e6 cells). The succinct verifier —
    the one that accepts that real proof with these same chip programs — accepts the GPU proof."""
    counts = {k: max(v // 16, 8) for k, v in RT.REFERENCE_COMPRESS_HEIGHTS.items()}
    tabs, pv = RT.generate(counts, seed=11)
    m = R.compress_machine()
    L, lsh = 17, 16
    jp = api.JaggedProver(L, lsh, 32, 2)
    dev = _device_chips(api, m, tabs)
    commit, prep = jp.commit_multilinears([d[3] for d in dev])
    ch, v = api.DuplexChallenger(), orc.Challenger()
    ch.observe(commit)
    v.observe(commit)
    proof = api.prove_shard(dev, pv, prep, L, lsh, 32, ch)
    assert orc.shard_verify(_shapes_only(m), commit, proof, L, lsh, v, 2, 124, 16) == 0
    assert np.array_equal(v.state(), ch.state())
    # a proof of a trace with one wrong Poseidon2 cell is rejected
    bad = tabs["Poseidon2WideDeg3"][1].copy()
    bad[77, 140] ^= 1
    dev2 = [(a, i, api.ColMajor.from_row_major_host(bad) if a.name == "Poseidon2WideDeg3" else mm, p) for a, i, mm, p in dev]
    ch, v = api.DuplexChallenger(), orc.Challenger()
    ch.observe(commit)
    v.observe(commit)
    proof = api.prove_shard(dev2, pv, prep, L, lsh, 32, ch)
    assert orc.shard_verify(_shapes_only(m), commit, proof, L, lsh, v, 2, 124, 16) != 0


def test_proving_key_slot_and_pow_witness_injection(api):
    """The AirProver-shaped entry points: sp1hip_setup (commit the preprocessed traces -> pk + vk), sp1hip_vk_observe_into,
    sp1hip_prove_shard_with_pk. Same bytes as the stage-by-stage path; and a caller-supplied proof-of-work witness is used
    (the reference's `find_any` returns any valid one, this library the smallest): the proof changes, still verifies."""
    import struct
    counts = {"BaseAlu": 70, "ExtAlu": 90, "MemoryConst": 50, "MemoryVar": 40, "Poseidon2WideDeg3": 20, "PrefixSumChecks": 33, "Select": 100}
    L, lsh, batch, fri = 8, 7, 4, (1, 5, 4)
    tabs, pv = RT.generate(counts, seed=21)
    m = R.compress_machine()
    dev = _device_chips(api, m, tabs)
    pc_start = orc.to_monty(np.array([3, 1, 4], np.uint32))
    cum = orc.to_monty(np.arange(100, 114, dtype=np.uint32))
    pk = api.ProvingKey([d[3] for d in dev], L, lsh, batch, pc_start, cum, 0, *fri)
    got = pk.prove_shard(dev, pv)
    # the same through the stage-level API
    jp = api.JaggedProver(L, lsh, batch, fri[0])
    commit, prep = jp.commit_multilinears([d[3] for d in dev])
    assert np.array_equal(commit, pk.preprocessed_commit)
    ch = api.DuplexChallenger()
    pk.observe_into(ch)
    head = ch.clone()
    assert api.prove_shard(dev, pv, prep, L, lsh, batch, ch, *fri) == got
    # oracle verifier from the same transcript head: vk.observe_into = commit, pc_start, septic x/y, flag, 6 zeros
    v = orc.Challenger()
    v.observe(np.concatenate([commit, pc_start, cum, np.zeros(7, np.uint32)]))
    assert np.array_equal(v.state(), head.state())
    assert orc.shard_verify(_shapes_only(m), commit, got, L, lsh, v.clone(), *fri) == 0
    # an alternative witness for the first grind (LogUp-GKR, 12 bits): rebuild the transcript up to that point
    t = head.clone()
    t.observe(pv)
    n_pv = struct.unpack_from("<Q", got, 0)[0]
    main_commit = orc.to_monty(np.frombuffer(got, dtype=np.uint32, count=8, offset=8 + 4 * n_pv))
    t.observe(main_commit)
    t.observe(orc.to_monty(np.array([len(m)], np.uint32)))
    for a, _ in m:
        name = a.name.encode()
        t.observe(orc.to_monty(np.array([tabs[a.name][1].shape[0], len(name)] + list(name), np.uint32)))
    valid = [w for w in range(40000) if t.clone().check_witness(12, int(orc.to_monty(np.array([w], np.uint32))[0]))][:2]
    assert len(valid) == 2
    other = pk.prove_shard(dev, pv, pow_witnesses=[valid[1]])
    assert other != got and len(other) == len(got)
    assert orc.shard_verify(_shapes_only(m), commit, other, L, lsh, v.clone(), *fri) == 0
    assert pk.prove_shard(dev, pv, pow_witnesses=[valid[0]]) == got           # the smallest one is the default
    with pytest.raises(api._lib.Sp1HipError):
        pk.prove_shard(dev, pv, pow_witnesses=[valid[0] + 1 if valid[0] + 1 != valid[1] else valid[0] + 2])


def test_device_trace_generation_matches_the_host_traces(api):
    """`generate_trace_device` for the recursion chips (sp1hip_tracegen_recursion_*): from the event arrays alone the device
    writes, column-major, exactly the tables the host generator built row by row — including the padding rows (zeros;
    Poseidon2: the permutation trace of the zero state), and for Poseidon2Wide all 179 intermediate-state columns,
    whose values the machine's constraints and the oracle permutation pin."""
    counts = {"BaseAlu": 700, "ExtAlu": 300, "MemoryConst": 500, "MemoryVar": 333, "Poseidon2WideDeg3": 150, "PrefixSumChecks": 97,
              "Select": 410}
    tabs, _ = RT.generate(counts, seed=5)

    def events_of(name):
        main = tabs[name][1]
        k = counts[name]
        if name == "MemoryVar":
            return main[:k].reshape(2 * k, 4)
        if name == "PrefixSumChecks":                       # event: x1, x2[4], zero, one[4], acc[4], new_acc[4], field_acc, new_field_acc
            ev = np.zeros((k, 20), np.uint32)
            ev[:, 0:5], ev[:, 10:20] = main[:k, 0:5], main[:k, 5:15]
            ev[:, 6] = orc.to_monty(np.array([1], np.uint32))[0]
            return ev
        if name == "Poseidon2WideDeg3":                     # event: input[16] | output[16]
            return np.concatenate([main[:k, 0:16], main[:k, 163:179]], axis=1)
        return main[:k]

    for name in api.RECURSION_TRACEGEN:
        want = tabs[name][1]
        got = api.tracegen_recursion(name, np.ascontiguousarray(events_of(name)), want.shape[0]).to_row_major_host()
        assert np.array_equal(got, want), name
    # and a shard proven from device-generated traces is byte-identical to the one from host traces
    m = R.compress_machine()
    L, lsh, batch, fri = 11, 9, 32, (1, 5, 4)
    dev_host = _device_chips(api, m, tabs)
    dev_gen = [(a, i, api.tracegen_recursion(a.name, np.ascontiguousarray(events_of(a.name)), tabs[a.name][1].shape[0])
                if a.name in api.RECURSION_TRACEGEN else mm, p) for a, i, mm, p in dev_host]
    pk = api.ProvingKey([d[3] for d in dev_host], L, lsh, batch, log_blowup=fri[0], num_queries=fri[1], pow_bits=fri[2])
    _, pv = RT.generate(counts, seed=5)
    assert pk.prove_shard(dev_gen, pv) == pk.prove_shard(dev_host, pv)
