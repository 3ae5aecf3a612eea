"""CPU-only checks of the oracle's PCS half: RS encode vs a naive DFT, Merkle commit vs the pure-Python
hash, BaseFold prover -> verifier round trips (the reference's own test style:
/root/reference/slop/crates/basefold-prover/src/prover.rs:L288-L361,
/root/reference/slop/crates/merkle-tree/src/p3sync.rs:L240-L305), stacked interleave and challenger."""
import numpy as np
import pytest

import kb_py
import pyoracle as orc

P = kb_py.P


def M(x):
    return orc.to_monty(np.asarray(x, dtype=np.uint32))


def C(x):
    return orc.from_monty(x)


@pytest.mark.parametrize("log_n,log_blowup,w", [(0, 0, 1), (0, 2, 3), (1, 1, 2), (3, 2, 5), (5, 1, 4), (6, 2, 1)])
def test_rs_encode_matches_naive_dft(log_n, log_blowup, w):
    rng = np.random.default_rng(log_n * 7 + log_blowup)
    n, N = 1 << log_n, 1 << (log_n + log_blowup)
    m = rng.integers(0, P, (n, w))
    got = C(orc.rs_encode(M(m), log_blowup))
    g = kb_py.two_adic_generator(log_n + log_blowup)
    for j in range(N):
        k = kb_py.reverse_bits_len(j, log_n + log_blowup)
        x = pow(g, k, P)
        for c in range(w):
            want = sum(int(m[i, c]) * pow(x, i, P) for i in range(n)) % P
            assert int(got[j, c]) == want


def test_rs_encode_linearity_and_systematic_zero():
    rng = np.random.default_rng(5)
    a = rng.integers(0, P, (256, 3))
    b = rng.integers(0, P, (256, 3))
    ea, eb, es = (C(orc.rs_encode(M(x), 2)) for x in (a, b, (a + b) % P))
    assert np.array_equal((ea.astype(np.int64) + eb) % P, es)
    assert not orc.rs_encode(np.zeros((64, 2), np.uint32), 2).any()


def test_merkle_commit_matches_python_and_reference_test_shape():
    """[2^10 x 25] x 10 tensors is the reference's own test shape; here 2^4 x {25,3} keeps pure Python fast."""
    rng = np.random.default_rng(11)
    ts = [rng.integers(0, P, (16, w)) for w in (25, 3)]
    mt = orc.MerkleTree([M(t) for t in ts])
    layers = C(mt.layers())
    leaves = [kb_py.hash_felts([int(v) for t in ts for v in t[i]]) for i in range(16)]
    assert layers[:16].tolist() == leaves
    cur, off = leaves, 16
    while len(cur) > 1:
        cur = [kb_py.compress(cur[2 * i], cur[2 * i + 1]) for i in range(len(cur) // 2)]
        assert layers[off:off + len(cur)].tolist() == cur
        off += len(cur)
    want = kb_py.compress(cur[0], kb_py.hash_felts([4, 28]))
    assert C(mt.commit).tolist() == want
    idx = [3, 0, 15, 3, 8]
    paths = mt.paths(idx)
    vals = np.concatenate([M(t)[idx] for t in ts], axis=1)
    assert orc.merkle_verify(mt.commit, idx, vals, 4, mt.root(), paths) == 0
    assert orc.merkle_verify(mt.commit, [3, 0, 15, 2, 8], vals, 4, mt.root(), paths) != 0


def test_partial_lagrange_and_eval():
    rng = np.random.default_rng(13)
    pt = rng.integers(0, P, (3, 4))
    eq = C(orc.partial_lagrange(M(pt)))
    one = [1, 0, 0, 0]
    for i in range(8):
        acc = one
        for j in range(3):
            bit = (i >> (2 - j)) & 1
            x = pt[j].tolist()
            acc = kb_py.ext_mul(acc, x if bit else kb_py.ext_sub(one, x))
        assert eq[i].tolist() == acc
    mle = rng.integers(0, P, (8, 2))
    ev = C(orc.eval_mle(M(mle), M(pt)))
    for c in range(2):
        acc = [0, 0, 0, 0]
        for i in range(8):
            acc = kb_py.ext_add(acc, kb_py.ext_scale(eq[i].tolist(), int(mle[i, c])))
        assert ev[c].tolist() == acc


def test_fold_mle_and_codeword_commute_with_encoding():
    """RS-encoding commutes with even/odd folding: fold(encode(m)) == encode(fold(m))."""
    rng = np.random.default_rng(17)
    m = M(rng.integers(0, P, (32, 4)))           # an ext mle as [n][4]
    beta = M(rng.integers(0, P, 4))
    cw = orc.rs_encode(m, 2)
    lhs = orc.fold_even_odd(cw, beta)
    rhs = orc.rs_encode(orc.fold_mle(m, beta), 2)
    assert np.array_equal(lhs, rhs)


def test_challenger_duplex_semantics():
    ch = orc.Challenger()
    ch.observe(M([1, 2, 3]))
    s1 = ch.sample()
    # manual: state[0..3]=1,2,3 ; permute ; pop state[7]
    st = kb_py.permute([1, 2, 3] + [0] * 13)
    assert int(C(np.array([s1], np.uint32))[0]) == st[7]
    s2 = ch.sample()
    assert int(C(np.array([s2], np.uint32))[0]) == st[6]
    ch.observe(M([9]))                               # clears the output buffer
    s3 = ch.sample()
    st2 = kb_py.permute([9] + st[1:])
    assert int(C(np.array([s3], np.uint32))[0]) == st2[7]
    # eight observations trigger a duplexing on their own
    ch2 = orc.Challenger()
    ch2.observe(M(list(range(8))))
    assert ch2.state()[16] == 0 and ch2.state()[25] == 8
    # grind returns the smallest valid witness and leaves the challenger in the post-check state
    base = orc.Challenger()
    base.observe(M([5, 6, 7]))
    probe = base.clone()
    w = base.grind(6)
    wc = int(C(np.array([w], np.uint32))[0])
    for cand in range(wc):
        assert not probe.clone().check_witness(6, int(M([cand])[0]))
    assert probe.check_witness(6, w)
    assert np.array_equal(probe.state(), base.state())


def _prove_and_verify(log_n, widths_per_round, log_blowup, nq, pow_bits, seed):
    rng = np.random.default_rng(seed)
    rounds = [orc.CommittedRound([M(rng.integers(0, P, (1 << log_n, w))) for w in ws], log_blowup)
              for ws in widths_per_round]
    ch = orc.Challenger()
    for r in rounds:
        ch.observe(r.commit)
    point = ch.sample_point(log_n)
    claims = [[orc.eval_mle(m, point) for m in r.mles] for r in rounds]
    verifier_ch = ch.clone()
    blob = orc.basefold_prove(point, rounds, claims, ch, log_blowup, nq, pow_bits)
    per_round = [np.concatenate(c) for c in claims]
    commits = [r.commit for r in rounds]
    rc = orc.basefold_verify(commits, point, per_round, blob, verifier_ch.clone(), log_blowup, nq, pow_bits)
    return rc, blob, commits, point, per_round, verifier_ch


@pytest.mark.parametrize("log_n,widths,lb", [(6, [[16, 10, 14], [20, 78, 34], [10, 10]], 1), (5, [[3]], 2),
                                              (1, [[2, 1]], 2)])
def test_basefold_roundtrip(log_n, widths, lb):
    rc, blob, commits, point, claims, vch = _prove_and_verify(log_n, widths, lb, 20, 6, 23 + log_n)
    assert rc == 0
    # tamper: a flipped byte anywhere in the query openings or a wrong claim must be rejected
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 1
    assert orc.basefold_verify(commits, point, claims, bytes(bad), vch.clone(), lb, 20, 6) != 0
    wrong = [c.copy() for c in claims]
    wrong[0][0, 0] = (int(wrong[0][0, 0]) + 1) % P
    assert orc.basefold_verify(commits, point, wrong, blob, vch.clone(), lb, 20, 6) != 0
    assert orc.basefold_verify(commits, point, claims, blob[:-3], vch.clone(), lb, 20, 6) == -1


def test_basefold_proof_is_deterministic():
    a = _prove_and_verify(4, [[5, 2]], 2, 8, 4, 99)[1]
    b = _prove_and_verify(4, [[5, 2]], 2, 8, 4, 99)[1]
    assert a == b


def test_interleave_fixed_rate():
    rng = np.random.default_rng(31)
    tabs = [rng.integers(0, P, (r, c)).astype(np.uint32) for r, c in ((8, 3), (4, 1), (16, 2), (2, 5))]
    lsh, bs = 3, 2
    got = orc.interleave(tabs, bs, lsh)
    dense = np.concatenate([t.T.reshape(-1) for t in tabs])
    H = 1 << lsh
    pad = (-len(dense)) % H
    dense = np.concatenate([dense, np.zeros(pad, np.uint32)])
    cols = dense.reshape(-1, H)
    want = [cols[i:i + bs].T for i in range(0, len(cols), bs)]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_babybear_oracle_field_ntt_and_commit_conventions():
    """The BabyBear oracle (oracle/bb_commit.hpp): published field constants (two-adic generator of order 2^27 = 440564289), the
    RS encode is the DFT of the zero-padded coefficients in bit-reversed order (checked against a naive evaluation), and the
    commitment chain is compress(root, hash([log_height, total_width])) over a leaf-first tree."""
    import ctypes as C
    lib = orc.lib()
    lib.orc_bb_two_adic_generator.restype = C.c_uint32
    P = orc.BB_P
    assert P == (1 << 31) - (1 << 27) + 1 and lib.orc_bb_two_adic_generator(27) == 440564289
    g = lib.orc_bb_two_adic_generator(7)
    assert pow(g, 128, P) == 1 and pow(g, 64, P) != 1
    ms = [orc.bb_random_felts((64, 5), 1), orc.bb_random_felts((64, 3), 2)]
    commit, cws, tree = orc.bb_commit_mles(ms, 1, True, True)
    co = orc.bb_from_monty(ms[1][:, 2]).astype(object)
    cw = orc.bb_from_monty(cws[1][:, 2])
    rev = lambda x: int(format(x, "07b")[::-1], 2)
    for j in range(0, 128, 5):
        assert sum(int(co[i]) * pow(g, (rev(j) * i) % 128, P) for i in range(64)) % P == int(cw[j])
    # leaf 3 = sponge over row 3 of both codewords; parents = compress; commitment = compress(root, hash([7, 8]))
    perm = lambda s: orc.bb_permute(np.array(s, np.uint32).reshape(1, 16))[0]
    row = list(cws[0][3]) + list(cws[1][3])
    st = perm(row[:8] + [0] * 8)
    assert np.array_equal(st[:8], tree[3])
    assert np.array_equal(perm(list(tree[0]) + list(tree[1]))[:8], tree[128])
    meta = perm(list(orc.bb_to_monty(np.array([7, 8], np.uint32))) + [0] * 14)[:8]
    assert np.array_equal(perm(list(tree[-1]) + list(meta))[:8], commit)
    # the zero state is not a fixed point and the permutation is not KoalaBear's
    assert orc.bb_permute(np.zeros((1, 16), np.uint32)).any()
