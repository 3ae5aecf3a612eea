"""GPU parity (-m gpu) of one whole shard proof: sp1hip_prove_shard's bincode(ShardProof) and final transcript state
equal the oracle's prove_shard_with_data byte for byte, and the oracle's restated verify_shard — with every
chip-dependent check — accepts it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from shard_chips import make_shard_chips, preprocessed_round  # noqa: E402

LB, NQ, PW = 1, 5, 4


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


@pytest.mark.parametrize("n_tuples,L,lsh,batch,with_empty,dup", [
    (4, 3, 2, 2, False, 2),
    (5, 4, 3, 3, True, 3),
    (3, 5, 2, 4, False, 1),
    (200, 10, 6, 4, True, 3),
])
def test_shard_proof_matches_oracle(api, n_tuples, L, lsh, batch, with_empty, dup):
    chips, publics = make_shard_chips(n_tuples, 20 + L, with_empty, dup)
    o_prep = preprocessed_round(chips, L, lsh, batch, LB)
    jp = api.JaggedProver(L, lsh, batch, LB)
    dev = []
    for air, inter, main, prep in chips:
        d_main = api.ColMajor.from_row_major_host(main) if main.shape[0] else None
        d_prep = api.ColMajor.from_row_major_host(prep) if prep is not None and prep.shape[0] else None
        dev.append((air, inter, d_main, d_prep))
    g_commit, g_prep = jp.commit_multilinears([d[3] for d in dev if d[3] is not None])
    assert np.array_equal(g_commit, o_prep.commit)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    o_ch.observe(o_prep.commit)
    g_ch.observe(g_commit)
    v_ch = o_ch.clone()
    want = orc.shard_prove(chips, publics, o_prep, L, lsh, batch, o_ch, LB, NQ, PW)
    got = api.prove_shard(dev, publics, g_prep, L, lsh, batch, g_ch, LB, NQ, PW)
    assert len(got) == len(want)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    assert orc.shard_verify(chips, g_commit, got, L, lsh, v_ch, LB, NQ, PW) == 0


def test_shard_proof_with_split_persistent_leaf_hash(api, monkeypatch):
    """The commit's leaf hash in its split form (one launch per stacked batch, SP1HIP_COMMIT_OVERLAP=1: the form large
    commitments take) as a persistent grid of 3 workgroups (SP1HIP_LEAF_WGS: the rows are then walked by a grid-stride loop)."""
    monkeypatch.setenv("SP1HIP_COMMIT_OVERLAP", "1")
    monkeypatch.setenv("SP1HIP_LEAF_WGS", "3")
    test_shard_proof_matches_oracle(api, 200, 10, 6, 4, True, 3)


@pytest.mark.parametrize("scale_log2", [3, 0])
def test_large_shard_proof_is_accepted_by_the_pinned_verifier(api, scale_log2):
    """Size-independent parity at (and near) the full core-shard size: the verifier is succinct, so the oracle's
    restated verify_shard — the one that accepts the reference's real ShardProof — can check a proof of 4.0e8 trace
    cells (scale 0; 6.3e6 at scale 3) produced by sp1hip_prove_shard with the production parameters (blowup 4,
    124 queries, 16-bit PoW): every Merkle opening, fold, sumcheck round, lookup balance and constraint evaluation."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bench"))
    from synthetic_shard import build_shard
    L, lsh = 22 - scale_log2, 21 - scale_log2
    area_target = ((1 << 28) + (1 << 27)) >> (2 * scale_log2)
    chips, prep_prep, shapes, area = build_shard(L, lsh, area_target)
    jp = api.JaggedProver(L, lsh, 32, 2)
    prep_commit, prep_data = jp.commit_multilinears([prep_prep])
    ch = api.DuplexChallenger()
    ch.observe(prep_commit)
    v_ch = orc.Challenger()
    v_ch.observe(prep_commit)
    proof = api.prove_shard(chips, [], prep_data, L, lsh, 32, ch)
    shapes_only = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
                   for a, i, _, _ in chips]
    assert orc.shard_verify(shapes_only, prep_commit, proof, L, lsh, v_ch, 2, 124, 16) == 0
    assert np.array_equal(v_ch.state(), ch.state())
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    v2 = orc.Challenger()
    v2.observe(prep_commit)
    assert orc.shard_verify(shapes_only, prep_commit, bytes(bad), L, lsh, v2, 2, 124, 16) != 0


def test_concurrent_shard_provers_give_identical_proofs(api):
    """The library is re-entrant per stream: three host threads proving the same shard on three streams at once
    (their kernels interleave on the GPU) each return the bytes of the sequential proof."""
    import os
    import sys
    import threading
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bench"))
    from synthetic_shard import build_shard
    L, lsh = 18, 17
    chips, prep_prep, shapes, area = build_shard(L, lsh, ((1 << 28) + (1 << 27)) >> 8)
    jp = api.JaggedProver(L, lsh, 32, 2)
    prep_commit, prep_data = jp.commit_multilinears([prep_prep])

    def prove(stream):
        ch = api.DuplexChallenger()
        ch.observe(prep_commit)
        with torch.cuda.stream(stream):
            return api.prove_shard(chips, [], prep_data, L, lsh, 32, ch, stream=stream)

    ref = prove(torch.cuda.current_stream())
    torch.cuda.synchronize()
    out = {}

    def worker(k, s):
        out[k] = [prove(s) for _ in range(3)]

    threads = [threading.Thread(target=worker, args=(k, torch.cuda.Stream())) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert all(p == ref for k in out for p in out[k]) and len(out) == 3


def test_stream_release_between_generations_of_caller_streams(api):
    """A caller stream owns helper streams (the commit's encode stream, the zerocheck's fork streams), events and cached buffers:
    `sp1hip_stream_release` hands them back before the stream goes away, and the next generation of streams — which may reuse
    the handle values — proves the same bytes."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bench"))
    from synthetic_shard import build_shard
    from sp1_amd import _lib
    L, lsh = 16, 15
    chips, prep_prep, shapes, area = build_shard(L, lsh, ((1 << 28) + (1 << 27)) >> 10)
    jp = api.JaggedProver(L, lsh, 32, 2)
    prep_commit, prep_data = jp.commit_multilinears([prep_prep])

    def prove(stream):
        ch = api.DuplexChallenger()
        ch.observe(prep_commit)
        with torch.cuda.stream(stream):
            return api.prove_shard(chips, [], prep_data, L, lsh, 32, ch, stream=stream)

    ref = prove(torch.cuda.current_stream())
    for generation in range(4):
        s = torch.cuda.Stream()
        assert prove(s) == ref and prove(s) == ref
        _lib.check(_lib.load().sp1hip_stream_release(s.cuda_stream))
        del s
    torch.cuda.synchronize()
    with pytest.raises(_lib.Sp1HipError):
        _lib.check(_lib.load().sp1hip_stream_release(None))        # the default stream is never released


def test_prove_shard_size_protocol_and_transcript_rollback(api):
    chips, publics = make_shard_chips(4, 23)
    L, lsh, batch = 3, 2, 2
    jp = api.JaggedProver(L, lsh, batch, LB)
    dev = [(a, i, api.ColMajor.from_row_major_host(m) if m.shape[0] else None,
            api.ColMajor.from_row_major_host(p) if p is not None and p.shape[0] else None) for a, i, m, p in chips]
    _, g_prep = jp.commit_multilinears([d[3] for d in dev if d[3] is not None])
    ch = api.DuplexChallenger()
    before = ch.state()
    # a corrupted constraint program is rejected by the zerocheck stage: the caller's transcript must be untouched
    bad = list(dev)
    air = bad[0][0]
    import copy
    air2 = copy.deepcopy(air)
    air2.instrs[0] = (0, 99, 0)                      # LOAD_MAIN column 99
    bad[0] = (air2, bad[0][1], bad[0][2], bad[0][3])
    with pytest.raises(api._lib.Sp1HipError):
        api.prove_shard(bad, publics, g_prep, L, lsh, batch, ch, LB, NQ, PW)
    assert np.array_equal(ch.state(), before)
    good = api.prove_shard(dev, publics, g_prep, L, lsh, batch, ch, LB, NQ, PW)
    assert len(good) > 1000 and not np.array_equal(ch.state(), before)
