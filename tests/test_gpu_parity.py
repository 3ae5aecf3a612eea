"""GPU parity tests (-m gpu): every HIP entry point of the C ABI against the CPU oracle on the same
seeded inputs, bit-exact (KoalaBear is integer arithmetic — no tolerance anywhere).

Shapes follow the reference's own tests: Merkle [2^10 x 25] x 10 tensors with 5 openings
(/root/reference/slop/crates/merkle-tree/src/p3sync.rs:L240-L305), RS encode log sizes 1..15 batch 16
(/root/reference/sp1-gpu/crates/basefold/src/encoder.rs:L164-L211), BaseFold widths
[16,10,14],[20,78,34],[10,10] (/root/reference/slop/crates/basefold-prover/src/prover.rs:L288-L361).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import kb_py  # noqa: E402
import pyoracle as orc  # noqa: E402

P = kb_py.P


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def col_major_host(t):
    return t.to_row_major_host()


def test_monty_roundtrip_and_permute(api):
    rng = np.random.default_rng(1)
    canon = rng.integers(0, P, 4096).astype(np.uint32)
    d = api.to_device(canon)
    api.check(api._L().sp1hip_to_monty(api._dptr(d), canon.size, api._stream_ptr()))
    assert np.array_equal(api.to_host(d), orc.to_monty(canon))
    api.check(api._L().sp1hip_from_monty(api._dptr(d), canon.size, api._stream_ptr()))
    assert np.array_equal(api.to_host(d), canon)
    states = orc.random_felts((1000, 16), 3)
    d = api.to_device(states)
    api.check(api._L().sp1hip_poseidon2_permute(api._dptr(d), 1000, api._stream_ptr()))
    got = api.to_host(d, (1000, 16))
    for i in (0, 1, 17, 999):
        assert np.array_equal(got[i], orc.permute(states[i]))
    zero = api.to_device(np.zeros(16, np.uint32))
    api.check(api._L().sp1hip_poseidon2_permute(api._dptr(zero), 1, api._stream_ptr()))
    assert orc.from_monty(api.to_host(zero)).tolist() == kb_py.KAT_PERM_ZERO


def test_poseidon2_formulations_agree_on_a_quarter_billion_states(api):
    """The production permutation (exact fp64 linear layer + signed, correction-free S-boxes) against the all-integer
    formulation on 2^28 states = as many permutations as two BASELINE commit steps: random words, plus the edge
    patterns that stress the lazy ranges (all zero, all p - 1, single non-zero lanes, alternating 0 / p - 1).
    Both are also pinned to the oracle on the first states."""
    L = api._L()
    n = 1 << 24
    edge = np.zeros((64, 16), np.uint32)
    edge[1] = P - 1
    for k in range(16):
        edge[2 + k, k] = P - 1
        edge[18 + k, k] = 1
    edge[34, ::2] = P - 1
    edge[35, 1::2] = P - 1
    edge[36:] = np.random.default_rng(7).integers(P - 64, P, (28, 16))
    gen = torch.Generator(device="cuda")
    for rep in range(16):                                   # 16 x 2^24 states
        gen.manual_seed(100 + rep)
        a = torch.randint(0, P, (n * 16,), dtype=torch.int32, device="cuda", generator=gen)
        a[:edge.size] = torch.from_numpy(edge.astype(np.int32).reshape(-1)).cuda()
        b = a.clone()
        api.check(L.sp1hip_poseidon2_permute(api._dptr(a), n, api._stream_ptr()))
        api.check(L.sp1hip_poseidon2_permute_integer_form(api._dptr(b), n, api._stream_ptr()))
        assert torch.equal(a, b), "formulations differ (rep %d)" % rep
        if rep == 0:
            got = a[:edge.size].cpu().numpy().astype(np.uint32).reshape(64, 16)
            for i in range(64):
                assert np.array_equal(got[i], orc.permute(edge[i])), i
        assert int(a.max()) < P and int(a.min()) >= 0


def test_transpose_roundtrip(api):
    for shape in ((1, 1), (5, 3), (33, 65), (1000, 25), (4096, 7)):
        a = orc.random_felts(shape, 5)
        t = api.ColMajor.from_row_major_host(a)
        assert np.array_equal(api.to_host(t.words, (shape[1], shape[0])), a.T)
        assert np.array_equal(t.to_row_major_host(), a)


@pytest.mark.parametrize("lg_n", list(range(0, 16)))
def test_rs_encode_matches_oracle(api, lg_n):
    """log sizes 0..15, batch 16, blowup 1 and 2 (the reference's GPU-vs-CPU test sweeps 1..15 x 16)."""
    for lb in (1, 2):
        if lg_n + lb > 16:
            continue
        m = orc.random_felts((1 << lg_n, 16), 100 + lg_n)
        want = orc.rs_encode(m, lb)
        got = api.DftEncoder(lb).encode_batch([api.ColMajor.from_row_major_host(m)])[0]
        assert np.array_equal(got.to_row_major_host(), want), (lg_n, lb)


@pytest.mark.parametrize("lg_n,lb,w", [(17, 2, 3), (18, 2, 2), (20, 2, 1), (21, 2, 1), (20, 0, 1), (22, 2, 1), (22, 1, 2)])
def test_rs_encode_large_multi_pass(api, lg_n, lb, w):
    """Two- and three-pass plans up to the two-adicity limit 2^24 (core shards encode 2^21 -> 2^23)."""
    m = orc.random_felts((1 << lg_n, w), 7)
    want = orc.rs_encode(m, lb)
    got = api.DftEncoder(lb).encode_batch([api.ColMajor.from_row_major_host(m)])[0]
    assert np.array_equal(got.to_row_major_host(), want)


def test_rs_encode_rejects_bad_sizes(api):
    m = api.ColMajor(api.device_words(4), 4, 1)
    with pytest.raises(api._lib.Sp1HipError):
        api.DftEncoder(23).encode_batch([m])          # exceeds two-adicity
    out = api.device_words(16)
    st = api._L().sp1hip_rs_encode_batch(api._dptr(out), api._dptr(out), 2, 2, 1, api._stream_ptr())
    assert st == -1 and b"alias" in api._L().sp1hip_last_error()


@pytest.mark.parametrize("height,widths", [(1 << 10, [25] * 10), (1, [3]), (2, [8]), (256, [1]), (1 << 12, [7, 9, 16, 1]),
                                            (1 << 13, [32, 32, 32]), (64, [200])])
def test_merkle_commit_and_openings(api, height, widths):
    ts = [orc.random_felts((height, w), 40 + i) for i, w in enumerate(widths)]
    want = orc.MerkleTree(ts)
    tcs = api.MerkleTcsProver()
    d_ts = [api.ColMajor.from_row_major_host(t) for t in ts]
    commit, data = tcs.commit_tensors(d_ts)
    assert np.array_equal(commit, want.commit)
    assert np.array_equal(data.root, want.root())
    assert np.array_equal(api.to_host(data.tree, (2 * height - 1, 8)), want.layers())
    rng = np.random.default_rng(9)
    idx = rng.integers(0, height, 5).tolist() + [0, height - 1]
    vals = tcs.compute_openings_at_indices(d_ts, idx)
    assert np.array_equal(vals, np.concatenate([t[idx] for t in ts], axis=1))
    proof = tcs.prove_openings_at_indices(data, idx)
    assert np.array_equal(proof["paths"], want.paths(idx))
    lg = height.bit_length() - 1
    assert orc.merkle_verify(commit, idx, vals, lg, proof["merkle_root"], proof["paths"]) == 0


def test_golden_leaves_hash_like_the_reference(api):
    """Rows opened in the reference's real proof, hashed by the HIP leaf kernel, must walk their
    stored Merkle paths to the stored roots (Poseidon2/sponge on the GPU == reference)."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "kb_shrink_basefold.npz"))
    q = gold["query_indices"].astype(np.uint64)
    tcs = api.MerkleTcsProver()
    for name, log_h, shift in (("comp0", 22, 0), ("comp1", 22, 0), ("round00", 21, 1), ("round19", 2, 20)):
        vals = orc.to_monty(gold[name + "_values"])
        rows = 16                                         # pad the 12 rows to a power of two
        padded = np.zeros((rows, vals.shape[1]), np.uint32)
        padded[:vals.shape[0]] = vals
        _, data = tcs.commit_tensors([api.ColMajor.from_row_major_host(padded)])
        leaves = api.to_host(data.tree, (2 * rows - 1, 8))[:vals.shape[0]]
        paths = orc.to_monty(gold[name + "_paths"])
        root = orc.to_monty(gold[name + "_root"])
        for k in range(vals.shape[0]):
            node, idx = leaves[k], int(q[k]) >> shift
            for sib in paths[k]:
                node = orc.compress(node, sib) if idx & 1 == 0 else orc.compress(sib, node)
                idx >>= 1
            assert np.array_equal(node, root), (name, k)


def _ext_soa(a):
    return np.ascontiguousarray(a.T)     # [n][4] -> [4][n]


def test_basefold_kernels(api):
    L = api._L()
    s = api._stream_ptr()
    rng = np.random.default_rng(21)
    # partial_lagrange for dims 0..13
    for dim in (0, 1, 2, 5, 13):
        pt = orc.random_felts((dim, 4), 60 + dim)
        out = api.device_words(4 << dim)
        api.check(L.sp1hip_partial_lagrange(api._ext_array(pt), dim, api._dptr(out), s))
        assert np.array_equal(api.to_host(out, (4, 1 << dim)).T, orc.partial_lagrange(pt)), dim
    # batch + column evaluations over a multi-tensor message with ragged widths
    lg = 11
    ts = [orc.random_felts((1 << lg, w), 70 + i) for i, w in enumerate((5, 32, 1, 10))]
    d_ts = [api.ColMajor.from_row_major_host(t) for t in ts]
    tw = sum(t.shape[1] for t in ts)
    coeffs = orc.random_felts((tw, 4), 77)
    out = api.device_words(4 << lg)
    api.check(L.sp1hip_basefold_batch(api._tensor_array(d_ts), len(d_ts), lg, api._dptr(api.to_device(coeffs)),
                                      api._dptr(out), s))
    allcols = np.concatenate(ts, axis=1)
    want = np.zeros((1 << lg, 4), np.uint32)
    got = api.to_host(out, (4, 1 << lg)).T
    for r in (0, 1, 1000, (1 << lg) - 1):
        acc = [0, 0, 0, 0]
        for g in range(tw):
            acc = kb_py.ext_add(acc, kb_py.ext_scale(orc.from_monty(coeffs[g]).tolist(), int(orc.from_monty(allcols[r, g:g + 1])[0])))
        assert orc.from_monty(got[r]).tolist() == acc
    pt = orc.random_felts((lg, 4), 78)
    claims = api.BasefoldProver().evaluate_mles(d_ts, pt)
    assert np.array_equal(claims, np.concatenate([orc.eval_mle(t, pt) for t in ts]))
    # folds
    for lg_n in (1, 2, 3, 10, 15):
        cw = orc.random_felts((1 << lg_n, 4), 80 + lg_n)
        beta = orc.random_felts((4,), 81)
        o1 = api.device_words(2 << lg_n)
        d_cw = api.to_device(_ext_soa(cw))
        api.check(L.sp1hip_fold_even_odd(api._dptr(d_cw), lg_n, api._ext(beta), api._dptr(o1), s))
        assert np.array_equal(api.to_host(o1, (4, 1 << (lg_n - 1))).T, orc.fold_even_odd(cw, beta)), lg_n
        api.check(L.sp1hip_fold_mle(api._dptr(d_cw), lg_n, api._ext(beta), api._dptr(o1), s))
        assert np.array_equal(api.to_host(o1, (4, 1 << (lg_n - 1))).T, orc.fold_mle(cw, beta)), lg_n
        # fixed_at_zero = eval of the even entries at a (lg_n - 1)-point
        pt = orc.random_felts((lg_n - 1, 4), 82)
        eq = orc.partial_lagrange(pt)
        o4 = api.device_words(4)
        api.check(L.sp1hip_ext_fixed_at_zero(api._dptr(d_cw), lg_n, api._dptr(api.to_device(_ext_soa(eq))), api._dptr(o4), s))
        acc = [0, 0, 0, 0]
        ce, cq = orc.from_monty(cw), orc.from_monty(eq)
        for i in range(1 << (lg_n - 1)):
            acc = kb_py.ext_add(acc, kb_py.ext_mul(cq[i].tolist(), ce[2 * i].tolist()))
        assert orc.from_monty(api.to_host(o4)).tolist() == acc, lg_n


def test_challenger_matches_oracle(api):
    rng = np.random.default_rng(33)
    a, b = api.DuplexChallenger(), orc.Challenger()
    for step in range(200):
        op = rng.integers(0, 4)
        if op == 0:
            xs = orc.random_felts((int(rng.integers(1, 12)),), 1000 + step)
            a.observe(xs)
            b.observe(xs)
        elif op == 1:
            assert a.sample() == b.sample()
        elif op == 2:
            assert np.array_equal(a.sample_ext_element(), b.sample_ext())
        else:
            bits = int(rng.integers(1, 24))
            assert a.sample_bits(bits) == b.sample_bits(bits)
        assert np.array_equal(a.state(), b.state())
    # grinding: host path (<= 8 bits) and GPU path, smallest witness, same post-state
    for bits in (5, 8, 12, 16):
        a.observe(orc.random_felts((3,), bits))
        b.observe(orc.random_felts((3,), bits))
        assert a.grind(bits) == b.grind(bits), bits
        assert np.array_equal(a.state(), b.state())
    with pytest.raises(api._lib.Sp1HipError):
        a.observe(np.array([P], np.uint32))              # non-reduced word


def test_gpu_grind_at_the_reference_pow_point(api):
    """Replay the reference's real transcript up to its 16-bit query-phase grind, then grind on the GPU:
    the kernel's smallest witness is accepted, is <= the reference's (any-valid) witness, and is the
    oracle's; the reference's own witness is accepted from the same state."""
    import transcript_tape as tt
    k = tt.pow_op_index(16)
    g, o = api.DuplexChallenger(), orc.Challenger()
    assert tt.replay(g, stop_before_op=k)[0] == k and tt.replay(o, stop_before_op=k)[0] == k
    ref_w = int(tt.TAPE["data"][tt.TAPE["ops"][k][2]])
    assert g.clone().check_witness(16, int(orc.to_monty(np.array([ref_w], np.uint32))[0]))
    w = g.grind(16)
    assert w == o.grind(16) and np.array_equal(g.state(), o.state())
    assert int(orc.from_monty(np.array([w], np.uint32))[0]) <= ref_w


def _prove_both(api, lg_n, widths_per_round, lb, nq, pow_bits, seed):
    mles = [[orc.random_felts((1 << lg_n, w), seed + 10 * r + i) for i, w in enumerate(ws)]
            for r, ws in enumerate(widths_per_round)]
    o_rounds = [orc.CommittedRound(ms, lb) for ms in mles]
    o_ch = orc.Challenger()
    for r in o_rounds:
        o_ch.observe(r.commit)
    o_pt = o_ch.sample_point(lg_n)
    o_claims = [[orc.eval_mle(m, o_pt) for m in ms] for ms in mles]
    o_blob = orc.basefold_prove(o_pt, o_rounds, o_claims, o_ch, lb, nq, pow_bits)

    prover = api.BasefoldProver(lb, nq, pow_bits)
    d_mles = [[api.ColMajor.from_row_major_host(m) for m in ms] for ms in mles]
    ch = api.DuplexChallenger()
    pds = []
    for r, ms in enumerate(d_mles):
        commit, pd = prover.commit_mles(ms)
        assert np.array_equal(commit, o_rounds[r].commit)
        for k in range(len(ms)):
            assert np.array_equal(pd.codeword(k).to_row_major_host(), o_rounds[r].codeword(k))
        assert np.array_equal(pd.tree(), o_rounds[r].layers())
        ch.observe(commit)
        pds.append(pd)
    pt = ch.sample_point(lg_n)
    assert np.array_equal(pt, o_pt)
    claims = prover.evaluate_mles([m for ms in d_mles for m in ms], pt)
    assert np.array_equal(claims, np.concatenate([c for rc in o_claims for c in rc]))
    v_ch = ch.clone()
    blob = prover.prove_trusted_mle_evaluations(pt, pds, claims, ch)
    assert blob == o_blob
    assert np.array_equal(ch.state(), o_ch.state())
    # and the oracle's verifier accepts the GPU proof
    ov = orc.Challenger()
    for r in o_rounds:
        ov.observe(r.commit)
    ov.sample_point(lg_n)
    per_round, k = [], 0
    for ws in widths_per_round:
        per_round.append(claims[k:k + sum(ws)])
        k += sum(ws)
    assert orc.basefold_verify([r.commit for r in o_rounds], pt, per_round, blob, ov, lb, nq, pow_bits) == 0
    return blob


@pytest.mark.parametrize("lg_n,widths,lb,nq,pw", [
    (10, [[16, 10, 14], [20, 78, 34], [10, 10]], 1, 94, 16),      # the reference test's widths + default_fri_config
    (12, [[32, 32], [7]], 2, 124, 16),                              # core parameters
    (1, [[2, 1]], 2, 8, 4),                                         # smallest instance
    (16, [[16, 10, 14]], 2, 124, 16),                               # the reference test's 2^16 variables
])
def test_basefold_proof_bytes_match_oracle(api, lg_n, widths, lb, nq, pw):
    _prove_both(api, lg_n, widths, lb, nq, pw, seed=500 + lg_n)


def test_basefold_buffer_too_small_leaves_transcript_untouched(api):
    import ctypes as C
    m = api.ColMajor.from_row_major_host(orc.random_felts((16, 2), 1))
    prover = api.BasefoldProver(2, 4, 2)
    commit, pd = prover.commit_mles([m])
    ch = api.DuplexChallenger()
    before = ch.state()
    n = C.c_size_t(10)
    buf = (C.c_uint8 * 10)()
    handles = (C.c_void_p * 1)(pd.h)
    st = api._L().sp1hip_basefold_prove(api._ext_array(np.zeros((4, 4))), 4, handles, 1, api._ext_array(np.zeros((2, 4))),
                                        2, prover.config, ch.h, buf, C.byref(n), api._stream_ptr())
    assert st == -6 and n.value > 10
    assert np.array_equal(ch.state(), before)


def test_full_size_properties(api):
    """BASELINE config 2 scale (2^20 rows): size-independent properties instead of the (slow) oracle:
    linearity of the encoder, fold(encode(m)) == encode(fold(m)), and oracle-verified Merkle paths of
    the full-size tree."""
    lg_n, lb = 20, 2
    L, s = api._L(), api._stream_ptr()
    a = orc.random_felts((4, 1 << lg_n), 1).T.copy()      # an ext mle [n][4]
    b = orc.random_felts((4, 1 << lg_n), 2).T.copy()
    ssum = ((orc.from_monty(a).astype(np.uint64) + orc.from_monty(b)) % P).astype(np.uint32)
    enc = api.DftEncoder(lb)
    da, db, dsum = (api.ColMajor(api.to_device(_ext_soa(x)), 1 << lg_n, 4) for x in (a, b, orc.to_monty(ssum)))
    ea, eb, es = enc.encode_batch([da, db, dsum])
    ha, hb, hs = (api.to_host(x.words) for x in (ea, eb, es))
    assert np.array_equal(((orc.from_monty(ha).astype(np.uint64) + orc.from_monty(hb)) % P).astype(np.uint32),
                          orc.from_monty(hs))
    beta = orc.random_felts((4,), 3)
    folded_cw = api.device_words(4 << (lg_n + lb - 1))
    api.check(L.sp1hip_fold_even_odd(api._dptr(ea.words), lg_n + lb, api._ext(beta), api._dptr(folded_cw), s))
    folded_m = api.device_words(4 << (lg_n - 1))
    api.check(L.sp1hip_fold_mle(api._dptr(da.words), lg_n, api._ext(beta), api._dptr(folded_m), s))
    enc2 = enc.encode_batch([api.ColMajor(folded_m, 1 << (lg_n - 1), 4)])[0]
    assert torch.equal(enc2.words, folded_cw)
    # full-size Merkle tree over the 4-column codeword: spot-check paths with the oracle's verifier
    tcs = api.MerkleTcsProver()
    commit, data = tcs.commit_tensors([ea])
    idx = [0, 1, (1 << 22) - 1, 123456, 3999999]
    vals = tcs.compute_openings_at_indices([ea], idx)
    proof = tcs.prove_openings_at_indices(data, idx)
    assert orc.merkle_verify(commit, idx, vals, lg_n + lb, proof["merkle_root"], proof["paths"]) == 0


def test_baseline_config2_commit_properties(api):
    """The bench workload itself (2^20 x 256 KoalaBear trace as 8 batches of 32, blowup 4): checksums of the full
    codeword that do not need the (slow) oracle — row 0 of the bit-reversed DFT is the evaluation at 1 = the column
    sum, row 1 the evaluation at -1 = the alternating sum, over all 256 columns; and oracle-verified Merkle openings
    (all 256 values of a row + 22-digest paths) of the full-size 8-tensor tree against the commitment."""
    lg_n, lb, W, B = 20, 2, 32, 8
    n = 1 << lg_n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    mles = [api.ColMajor(torch.randint(0, P, (W * n,), dtype=torch.int32, device="cuda", generator=gen), n, W) for _ in range(B)]
    enc = api.DftEncoder(lb)
    cws = enc.encode_batch(mles)
    Lf = api._L()
    for m, cw in zip(mles, cws):
        canon = m.words.clone()
        api.check(Lf.sp1hip_from_monty(api._dptr(canon), canon.numel(), api._stream_ptr()))
        cols = canon.view(W, n).to(torch.int64)
        total = cols.sum(dim=1) % P
        alt = (cols[:, 0::2].sum(dim=1) - cols[:, 1::2].sum(dim=1)) % P
        out = cw.words.view(W, n << lb)[:, :2].clone().contiguous()
        api.check(Lf.sp1hip_from_monty(api._dptr(out), out.numel(), api._stream_ptr()))
        assert torch.equal(out[:, 0].to(torch.int64), total) and torch.equal(out[:, 1].to(torch.int64), alt)
    tcs = api.MerkleTcsProver()
    commit, data = tcs.commit_tensors(cws)
    idx = [0, 1, (1 << 22) - 1, 2718281, 3141592]
    vals = tcs.compute_openings_at_indices(cws, idx)
    proof = tcs.prove_openings_at_indices(data, idx)
    assert vals.shape[-1] == W * B
    assert orc.merkle_verify(commit, idx, vals, lg_n + lb, proof["merkle_root"], proof["paths"]) == 0
    # the same commitment through the prover entry point the bench times
    commit2, _ = api.BasefoldProver(lb, 124, 16).commit_mles(mles)
    assert np.array_equal(commit, commit2)


@pytest.mark.parametrize("mode", ["1", "0"])
@pytest.mark.parametrize("lg_n,lb,widths", [
    (10, 2, [16, 8, 24]),          # splittable: every boundary on a multiple of the sponge rate
    (12, 1, [8, 8, 13]),           # ragged tail in the last tensor
    (9, 2, [32, 32, 32, 32, 32, 32]),
    (11, 2, [16, 10, 14]),         # parts {16}, {10 + 14}: a part closes where the row is a whole number of blocks
    (10, 2, [32, 32, 3]),          # a tail narrower than one block joins the part before it
    (8, 2, [8, 4]),                # ... which leaves a single part here -> single-launch path
    (9, 1, [5, 6, 7]),             # never block-aligned -> single-launch path
    (0, 2, [8, 8]),                # one-row tensors
])
def test_commit_mles_overlapped_encode_and_hash(api, monkeypatch, mode, lg_n, lb, widths):
    """commit_mles hashes tensor k on the caller's stream while tensor k + 1 is encoded on a side stream, carrying
    the 8 capacity words of every row's sponge across tensor boundaries (leaf_hash_part_kernel). Forced on ("1")
    and off ("0") at sizes the oracle handles: commitment, every codeword and the whole tree are the oracle's."""
    monkeypatch.setenv("SP1HIP_COMMIT_OVERLAP", mode)
    ms = [orc.random_felts((1 << lg_n, w), 900 + 7 * i + lg_n) for i, w in enumerate(widths)]
    o = orc.CommittedRound(ms, lb)
    prover = api.BasefoldProver(lb, 124, 16)
    d = [api.ColMajor.from_row_major_host(m) for m in ms]
    for _ in range(2):             # second pass re-uses the side stream, its events and recycled buffers
        commit, pd = prover.commit_mles(d)
        assert np.array_equal(commit, o.commit)
        for k in range(len(ms)):
            assert np.array_equal(pd.codeword(k).to_row_major_host(), o.codeword(k))
        assert np.array_equal(pd.tree(), o.layers())
        del pd


def test_commit_mles_split_plan_on_random_width_lists(api, monkeypatch):
    """leaf_hash_plan groups tensors into launches wherever the row so far is a whole number of sponge blocks. 40 random
    width lists (zero-width tensors, narrow tails, long unaligned runs): the overlapped commit and the single-launch
    commit agree on commitment and tree, and a sample of them is checked against the oracle as well."""
    rng = np.random.default_rng(2024)
    prover = api.BasefoldProver(1, 124, 16)
    for case in range(40):
        lg_n = int(rng.integers(0, 8))
        widths = [int(w) for w in rng.choice([0, 1, 3, 5, 8, 8, 8, 11, 16, 24, 40], size=int(rng.integers(1, 9)))]
        if sum(widths) == 0:
            widths.append(8)
        ms = [orc.random_felts((1 << lg_n, w), 7000 + 13 * case + i) if w else np.zeros((1 << lg_n, 0), np.uint32)
              for i, w in enumerate(widths)]
        d = [api.ColMajor.from_row_major_host(m) if m.shape[1] else
             api.ColMajor(torch.zeros(0, dtype=torch.int32, device="cuda"), 1 << lg_n, 0) for m in ms]
        monkeypatch.setenv("SP1HIP_COMMIT_OVERLAP", "1")
        c1, pd1 = prover.commit_mles(d)
        t1 = pd1.tree()
        monkeypatch.setenv("SP1HIP_COMMIT_OVERLAP", "0")
        c0, pd0 = prover.commit_mles(d)
        assert np.array_equal(c0, c1), (lg_n, widths)
        assert np.array_equal(pd0.tree(), t1), (lg_n, widths)
        if case % 8 == 0:
            assert np.array_equal(c1, orc.CommittedRound(ms, 1).commit), (lg_n, widths)


def _tables(shapes, seed):
    return [orc.random_felts(s, seed + i) if s[0] * s[1] else np.zeros(s, np.uint32) for i, s in enumerate(shapes)]


@pytest.mark.parametrize("shapes,lsh,batch", [
    ([(96, 3), (32, 1), (160, 2), (64, 5)], 6, 2),          # ragged heights, several batches, zero padding
    ([(64, 4)], 6, 4),                                       # exactly one full batch, no padding
    ([(64, 4), (64, 4)], 6, 4),                              # exact multiple of a full batch (overflow-buffer case)
    ([(40, 1)], 6, 32),                                      # less than one stacked column
    ([(1 << 12, 7), (3000, 13), (32, 30)], 10, 32),          # core-like: batch 32
])
def test_stacked_and_jagged_commit(api, shapes, lsh, batch):
    """a5/a7/a8: dense stacking of ragged chip tables, BaseFold commit of the batches, jagged wrapper."""
    lb = 2
    tables = _tables(shapes, 900)
    # oracle: interleave -> commit_mles -> wrapper
    o_batches = orc.interleave(tables, batch, lsh)
    o_round = orc.CommittedRound(o_batches, lb)
    area = sum(t.size for t in tables)
    H = 1 << lsh
    added = max(-(-area // H) * H, H) - area
    d_tables = [api.ColMajor.from_row_major_host(t) for t in tables]
    commit, sd, num_added = api.StackedPcsProver(lsh, batch, lb).commit_multilinears(d_tables)
    assert num_added == added
    assert [b.width for b in sd.batches] == [b.shape[1] for b in o_batches]
    assert np.array_equal(commit, o_round.commit)
    # jagged wrapper with a zero-row chip in the middle (counted, not committed)
    max_log_rows = max(lsh, max(int(s[0] - 1).bit_length() for s in shapes))
    with_empty = d_tables[:1] + [api.ColMajor(api.device_words(0), 0, 9)] + d_tables[1:]
    jcommit, jsd = api.JaggedProver(max_log_rows, lsh, batch, lb).commit_multilinears(with_empty)
    rows = [shapes[0][0], 0] + [s[0] for s in shapes[1:]]
    cols = [shapes[0][1], 9] + [s[1] for s in shapes[1:]]
    want = orc.jagged_commit_wrap(o_round.commit, rows, cols, added, max_log_rows)
    assert np.array_equal(jcommit, want)
    # the stacked batches open like any BaseFold commitment: prove + verify with the oracle
    prover = api.BasefoldProver(lb, 16, 8)
    ch = api.DuplexChallenger()
    ch.observe(commit)
    pt = ch.sample_point(lsh)
    claims = prover.evaluate_mles(sd.batches, pt)
    assert np.array_equal(claims, np.concatenate([orc.eval_mle(b, pt) for b in o_batches]))
    blob = prover.prove_trusted_mle_evaluations(pt, [sd.basefold], claims, ch)
    ov = orc.Challenger()
    ov.observe(commit)
    ov.sample_point(lsh)
    assert orc.basefold_verify([commit], pt, [claims], blob, ov, lb, 16, 8) == 0


def test_stacked_commit_of_nothing(api):
    """Empty message: one zero-width batch, padding = one stacked column (reference edge case)."""
    commit, sd, added = api.StackedPcsProver(4, 2, 1).commit_multilinears([])
    assert added == 16 and [b.width for b in sd.batches] == [0]
    o = orc.CommittedRound([np.zeros((16, 0), np.uint32)], 1)
    assert np.array_equal(commit, o.commit)


@pytest.mark.parametrize("W,lb", [(32, 2), (32, 1), (256, 2), (256, 1)])
def test_baseline_config2_commit_is_bit_exact_against_the_oracle(api, W, lb):
    """SURVEY §8(d) config 2, the whole contract: n = 2^20 rows, W in {32, 256} columns (32-column tensors, as the stacked
    prover commits them), log_blowup in {1, 2}, values from the documented generator (SplitMix64 seeds -> x mod p,
    `orc.random_felts`). The 8-word commitment always equals the oracle's; for W = 32 and for (W = 256, log_blowup 1) so do
    the FULL codeword and every layer of the Merkle tree (memcmp). Oracle time on the GPU box's 16 host threads: ~14 s for the four cases."""
    lg_n = 20
    ms = [orc.random_felts((1 << lg_n, 32), 4200 + 17 * k + lb) for k in range(W // 32)]
    o = orc.CommittedRound(ms, lb)
    d = [api.ColMajor.from_row_major_host(m) for m in ms]
    commit, pd = api.BasefoldProver(lb, 124, 16).commit_mles(d)
    assert np.array_equal(commit, o.commit)
    if W == 32 or lb == 1:                  # full memcmp: every codeword tensor and every layer of the Merkle tree
        for k in range(W // 32):            # (W = 256, log_blowup 1: 8 tensors of 2^21 x 32 words = 2.1 GB compared word for word)
            assert np.array_equal(pd.codeword(k).to_row_major_host(), o.codeword(k)), k
        assert np.array_equal(pd.tree(), o.layers())
    else:                                   # the root of the tree pins every leaf and every compression below it
        assert np.array_equal(pd.tree()[-1], o.layers()[-1])
    del pd
