#!/usr/bin/env python3
"""Replay the Fiat-Shamir transcript of the reference's one real shard proof and record it as a tape.

Source (read-only, only in the build container):
  /root/reference/sp1-gpu/crates/perf/recursion_records/shrink_input.bin  (see make_golden.py)

The whole (MachineVerifyingKey, ShardProof) pair is parsed (bincode; every length / shape field is
asserted) and the verifier's challenger calls are re-enacted in the order of
  vk.observe_into                 /root/reference/crates/hypercube/src/verifier/config.rs:L96-L112
  verify_shard                    /root/reference/crates/hypercube/src/verifier/shard.rs:L437-L488
  verify_logup_gkr                /root/reference/crates/hypercube/src/logup_gkr/verifier.rs:L102-L285
  verify_zerocheck                /root/reference/crates/hypercube/src/verifier/shard.rs:L304-L430
  jagged verify_trusted_evaluations /root/reference/slop/crates/jagged/src/verifier.rs:L109-L383
  jagged_evaluation               /root/reference/slop/crates/jagged/src/jagged_eval/sumcheck_eval.rs:L47-L80
  verify_untrusted_evaluation     /root/reference/slop/crates/multilinear/src/pcs.rs:L70-L90
  basefold verify_untrusted_evaluations / verify_mle_evaluations
                                  /root/reference/slop/crates/basefold/src/verifier.rs:L122-L237,L413-L430
  partially_verify_sumcheck_proof /root/reference/slop/crates/sumcheck/src/verifier.rs:L21-L95
with oracle/kb_py.py's pure-Python DuplexChallenger. The proof itself carries the answers, so the
replay is a known-answer test of the challenger (absorb order, duplexing rule, sample order,
sample_bits, check_witness) on ~20k absorbed words:
  * the 12-bit LogUp-GKR, 5-bit batch and 16-bit query grinding witnesses must pass check_witness,
  * every sumcheck's sampled point must equal the point stored in its PartialSumcheckProof
    (20 GKR rounds, zerocheck, jagged sumcheck, jagged eval),
  * each GKR round's claimed_sum must equal numerator_eval * lambda + denominator_eval and its final
    eval the eq-weighted product formula (pins the sampled lambda / last coordinates + ext arithmetic),
  * the zerocheck point must equal the LogUp evaluation point's continuation (trace point check),
  * the 20 BaseFold betas must equal the ones make_golden.py solved from the query openings,
  * the sampled query indices must equal the ones make_golden.py recovered from the Merkle paths.

Output tests/golden/kb_shrink_transcript.npz:
  ops  int32 [n, 4] = (opcode, arg, data offset, pinned)   data uint32 canonical words
    opcode 0 OBSERVE      arg = #words          data = the words
    opcode 1 SAMPLE       arg = #base samples   data = expected values (in sampling order)
    opcode 2 SAMPLE_BITS  arg = bits            data = expected value
    opcode 3 CHECK_WITNESS arg = bits           data = witness (must be accepted)
  basefold_proof_q12 = bincode(BasefoldProof) restricted to the first 12 of the 124 queries: the
    reference's own bytes (univariate messages, commitments, openings, Merkle paths, final_poly, both
    witnesses) with only the per-opening counts rewritten — what the oracle's BaseFold verifier is run
    on in tests/test_oracle_golden.py (num_queries = 12; the query indices are the first 12 sampled).
  jagged_tail = the reference's bytes of the rest of JaggedPcsProof (batch evaluations, jagged sumcheck,
    jagged-eval sumcheck, row/column counts, commitments, expected_eval, max_log_row_count, log_m);
    basefold_proof_q12 + jagged_tail is a complete JaggedPcsProof. jagged_start_op = tape index where
    JaggedPcsVerifier::verify_trusted_evaluations starts (first z_col sample); jagged_z_row = the
    zerocheck point; jagged_claims0/1 = the preprocessed / main column openings it is given.
  shard_head = the reference's bytes of the ShardProof from its first byte (public_values) up to the BasefoldProof:
    shard_head + basefold_proof_q12 + jagged_tail is a complete bincode(ShardProof) with 12 queries;
    shard_head[gkr_range[0]:gkr_range[1]] is bincode(LogupGkrProof). shard_start_op = tape index right after
    vk.observe_into (where verify_shard starts); gkr_start_op = tape index of the 12-bit grinding check (the
    first op of verify_logup_gkr).
  stack_point = the evaluation point of the stacked PCS (last log_stacking_height coordinates of the
    jagged sumcheck point), expected_eval = JaggedPcsProof.expected_eval.
  pinned = 1 when the expected value is read from / checked against the proof itself, 0 when it is
  only what the Python challenger produced (still reproduced by the other implementations).
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import kb_py as kb  # noqa: E402

SRC = "/root/reference/sp1-gpu/crates/perf/recursion_records/shrink_input.bin"
P = kb.P
GKR_GRINDING_BITS, BATCH_GRINDING_BITS, POW_BITS, NUM_QUERIES, LOG_BLOWUP = 12, 5, 16, 124, 2
NQ_KEEP = 12                                                 # queries kept in the trimmed BasefoldProof


class Reader:
    def __init__(self, buf, off=0):
        self.b, self.o = buf, off

    def u8(self):
        self.o += 1
        return self.b[self.o - 1]

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def felts(self, k):
        v = list(struct.unpack_from("<%dI" % k, self.b, self.o))
        self.o += 4 * k
        assert all(x < P for x in v), "non-canonical field element: parse is off"
        return v

    def ext(self):
        return self.felts(4)

    def exts(self, k):
        return [self.ext() for _ in range(k)]

    def vec_ext(self):
        return self.exts(self.u64())

    def string(self):
        n = self.u64()
        assert n < 64
        s = self.b[self.o:self.o + n].decode("ascii")
        self.o += n
        return s

    def tensor_ext(self, ndim):
        vals = self.vec_ext()
        dims = [self.u64() for _ in range(self.u64())]
        assert len(dims) == ndim and int(np.prod(dims)) == len(vals), (dims, len(vals))
        return vals, dims

    def sumcheck(self):
        polys = [self.vec_ext() for _ in range(self.u64())]
        claimed = self.ext()
        point = self.vec_ext()
        ev = self.ext()
        assert len(point) == len(polys)
        return dict(polys=polys, claimed_sum=claimed, point=point, eval=ev)

    def opening_trim(self, nq):
        """MerkleTreeOpeningAndProof (layout: make_golden.py). Returns the bincode of the same opening
        restricted to its first nq queries: the reference's own bytes, only the four counts rewritten."""
        n = self.u64()
        v0 = self.o
        self.o += 4 * n
        dims = [self.u64() for _ in range(self.u64())]
        assert len(dims) == 2 and dims[0] * dims[1] == n and dims[0] >= nq
        root = self.b[self.o:self.o + 32]
        self.o += 32
        log_h, width = self.u64(), self.u64()
        npath = self.u64()
        p0 = self.o
        self.o += 32 * npath
        pd = [self.u64() for _ in range(self.u64())]
        assert pd == [dims[0], log_h] and width == dims[1] and npath == dims[0] * log_h
        q = struct.pack
        return (q("<Q", nq * width) + self.b[v0:v0 + 4 * nq * width] + q("<QQQ", 2, nq, width) + root +
                q("<QQ", log_h, width) + q("<Q", nq * log_h) + self.b[p0:p0 + 32 * nq * log_h] +
                q("<QQQ", 2, nq, log_h))


class Tape:
    def __init__(self):
        self.ch = kb.Challenger()
        self.ops, self.data = [], []

    def _push(self, op, arg, words, pinned):
        self.ops.append((op, arg, len(self.data), pinned))
        self.data.extend(int(w) for w in words)

    def observe(self, words):
        words = list(words)
        if self.ops and self.ops[-1][0] == 0:               # merge adjacent observes
            op, arg, off, p = self.ops[-1]
            self.ops[-1] = (0, arg + len(words), off, 1)
            self.data.extend(words)
        else:
            self._push(0, len(words), words, 1)
        self.ch.observe_many(words)

    def barrier(self):
        """Do not merge the next observe into the previous one (so that a replay can stop exactly here)."""
        self.ops.append((0, 0, len(self.data), 1))

    def observe_exts(self, es):
        self.observe([w for e in es for w in e])

    def observe_var_exts(self, es):
        self.observe([len(es)] + [w for e in es for w in e])

    def sample_ext(self, expect=None, pinned=False):
        got = self.ch.sample_ext()
        if expect is not None:
            assert got == list(expect), "sampled challenge differs from the proof's"
        self._push(1, 4, got, int(expect is not None or pinned))
        return got

    def sample_bits(self, bits, expect=None):
        got = self.ch.sample_bits(bits)
        if expect is not None:
            assert got == expect, "sampled bits differ"
        self._push(2, bits, [got], int(expect is not None))
        return got

    def check_witness(self, bits, w):
        assert self.ch.check_witness(bits, w), "grinding witness rejected"
        self._push(3, bits, [w], 1)

    def sumcheck(self, sc, degree, mark_pinned=True):
        """partially_verify_sumcheck_proof: observe coefficients, sample alpha, ...; the sampled alphas
        inserted at the FRONT must equal the stored point."""
        n = len(sc["polys"])
        for k, poly in enumerate(sc["polys"]):
            assert len(poly) == degree + 1
            self.observe_exts(poly)
            self.sample_ext(expect=sc["point"][n - 1 - k] if mark_pinned else None)


def ext_sum(xs):
    acc = [0, 0, 0, 0]
    for x in xs:
        acc = kb.ext_add(acc, x)
    return acc


def eval_mle(vals, point):
    """vals: 2^n ext, point[0] is the most significant variable (slop Mle::eval_at convention)."""
    cur = list(vals)
    for z in point:
        half = len(cur) // 2
        cur = [kb.ext_add(cur[i], kb.ext_mul(z, kb.ext_sub(cur[half + i], cur[i]))) for i in range(half)]
    assert len(cur) == 1
    return cur[0]


def full_lagrange_eval(a, b):
    acc = kb.ext_from_base(1)
    one = kb.ext_from_base(1)
    for x, y in zip(a, b):
        xy = kb.ext_mul(x, y)
        t = kb.ext_add(kb.ext_sub(kb.ext_sub(one, x), y), kb.ext_add(xy, xy))    # 1 - x - y + 2xy
        acc = kb.ext_mul(acc, t)
    return acc


def main():
    b = open(SRC, "rb").read()
    gold = np.load(os.path.join(HERE, "kb_shrink_basefold.npz"))
    r = Reader(b)
    assert r.u64() == 1                                     # vks_and_proofs.len()
    pc_start = r.felts(3)
    gcs_x, gcs_y = r.felts(7), r.felts(7)
    pre_commit = r.felts(8)
    enable_untrusted = r.felts(1)[0]
    public_values = r.felts(r.u64())
    main_commit = r.felts(8)
    assert len(public_values) == 187 and r.o == 900

    # ---- LogupGkrProof --------------------------------------------------------------------------
    gkr_start = r.o
    numer, nd = r.tensor_ext(2)
    denom, dd = r.tensor_ext(2)
    assert nd == dd and nd[1] == 1
    rounds = []
    for _ in range(r.u64()):
        n0, n1, d0, d1 = r.ext(), r.ext(), r.ext(), r.ext()
        rounds.append(dict(n0=n0, n1=n1, d0=d0, d1=d1, sc=r.sumcheck()))
    logup_point = r.vec_ext()
    gkr_openings = []
    for _ in range(r.u64()):
        name = r.string()
        main_ev, _ = r.tensor_ext(1)
        prep_ev = r.tensor_ext(1)[0] if r.u8() else None
        gkr_openings.append((name, prep_ev, main_ev))
    gkr_witness = r.felts(1)[0]
    gkr_bytes = b[gkr_start:r.o]
    zerocheck = r.sumcheck()
    opened = []
    for _ in range(r.u64()):
        name = r.string()
        prep, mainv = r.vec_ext(), r.vec_ext()
        degree = r.felts(r.u64())
        opened.append((name, prep, mainv, degree))
    assert [o[0] for o in opened] == [g[0] for g in gkr_openings] == sorted(o[0] for o in opened)
    max_log_row_count = len(rounds) + 1
    assert all(len(o[3]) == max_log_row_count + 1 for o in opened)
    print("chips:", [(o[0], len(o[1]), len(o[2])) for o in opened], "max_log_row_count", max_log_row_count)

    # ---- JaggedPcsProof -------------------------------------------------------------------------
    bf_start = r.o
    n_uni = r.u64()
    uni = [r.exts(2) for _ in range(n_uni)]
    fri_commits = [r.felts(8) for _ in range(r.u64())]
    blob = bytearray(b[bf_start:r.o])
    for _ in range(2):                                       # component openings, then query-phase openings
        cnt = r.u64()
        blob += struct.pack("<Q", cnt)
        for _ in range(cnt):
            blob += r.opening_trim(NQ_KEEP)
    tail = r.o
    final_poly = r.ext()
    pow_witness, batch_witness = r.felts(1)[0], r.felts(1)[0]
    blob += b[tail:r.o]
    jagged_tail_start = r.o
    batch_evals = [r.tensor_ext(1)[0] for _ in range(r.u64())]
    jagged_sc = r.sumcheck()
    jagged_eval_sc = r.sumcheck()
    rc = [[(r.u64(), r.u64()) for _ in range(r.u64())] for _ in range(r.u64())]
    mt_commits = [r.felts(8) for _ in range(r.u64())]
    expected_eval = r.ext()
    assert r.u64() == max_log_row_count
    log_m = r.u64()
    jagged_tail = b[jagged_tail_start:r.o]
    print("parsed shard proof: %d bytes, log_m %d" % (r.o, log_m))
    assert pow_witness == int(gold["pow_witness"]) and batch_witness == int(gold["batch_witness"])

    # ======================= replay ==============================================================
    t = Tape()
    # vk.observe_into
    t.observe(pre_commit)
    t.observe(pc_start)
    t.observe(gcs_x)
    t.observe(gcs_y)
    t.observe([enable_untrusted])
    t.observe([0] * 6)
    t.barrier()
    shard_start_op = len(t.ops) - 1
    # verify_shard head
    t.observe(public_values)
    t.observe(main_commit)
    t.observe([len(opened)])
    for name, _, _, degree in opened:
        acc = 0
        for x in degree:
            acc = (x + 2 * acc) % P
        t.observe([acc, len(name)] + list(name.encode()))
    head = (list(t.ops), list(t.data), t.ch.__dict__.copy())

    # verify_logup_gkr: beta_seed_dim depends on the machine's widest interaction, which the proof
    # does not store — find it as the smallest value that makes the first GKR sumcheck point come out
    # (2; 3 draws the same number of sponge permutations before the next absorb and is indistinguishable).
    niv = (len(numer).bit_length() - 1) - 1                 # number_of_interaction_variables
    found = None
    for beta_seed_dim in range(1, 9):
        t = Tape()
        t.ops, t.data = list(head[0]), list(head[1])
        t.ch.state, t.ch.inp, t.ch.out = list(head[2]["state"]), list(head[2]["inp"]), list(head[2]["out"])
        try:
            t.check_witness(GKR_GRINDING_BITS, gkr_witness)
            t.sample_ext()                                  # alpha
            for _ in range(beta_seed_dim):
                t.sample_ext()
            t.sample_ext()                                  # pv_challenge
            t.observe_var_exts(numer)
            t.observe_var_exts(denom)
            eval_point = [t.sample_ext(pinned=True) for _ in range(niv + 1)]
            num_eval, den_eval = eval_mle(numer, eval_point), eval_mle(denom, eval_point)
            for i, rd in enumerate(rounds):
                lam = t.sample_ext(pinned=True)             # pinned by the two equations below
                assert rd["sc"]["claimed_sum"] == kb.ext_add(kb.ext_mul(num_eval, lam), den_eval), "gkr claim"
                assert len(rd["sc"]["polys"]) == i + niv + 1
                t.sumcheck(rd["sc"], 3)
                eq = full_lagrange_eval(rd["sc"]["point"], eval_point)
                nse = kb.ext_add(kb.ext_mul(rd["n0"], rd["d1"]), kb.ext_mul(rd["n1"], rd["d0"]))
                dse = kb.ext_mul(rd["d0"], rd["d1"])
                assert rd["sc"]["eval"] == kb.ext_mul(eq, kb.ext_add(kb.ext_mul(nse, lam), dse)), "gkr final eval"
                t.observe_exts([rd["n0"], rd["n1"], rd["d0"], rd["d1"]])
                last = t.sample_ext(pinned=True)            # enters the next round's claim
                eval_point = list(rd["sc"]["point"]) + [last]
                num_eval = kb.ext_add(rd["n0"], kb.ext_mul(kb.ext_sub(rd["n1"], rd["n0"]), last))
                den_eval = kb.ext_add(rd["d0"], kb.ext_mul(kb.ext_sub(rd["d1"], rd["d0"]), last))
            found = beta_seed_dim
            break
        except AssertionError as e:
            if beta_seed_dim == 8:
                raise
            last_err = e
    print("LogUp-GKR transcript OK: beta_seed_dim =", found, "interaction variables =", niv)
    trace_point = eval_point[niv:]
    assert trace_point == logup_point and len(trace_point) == max_log_row_count
    t.observe([len(opened)])
    for name, prep_ev, main_ev in gkr_openings:
        if prep_ev is not None:
            t.observe_var_exts(prep_ev)
        t.observe_var_exts(main_ev)

    # verify_zerocheck
    zerocheck_start_op = len(t.ops)
    t.sample_ext()                                          # alpha
    t.sample_ext()                                          # gkr_batch_open_challenge
    t.sample_ext()                                          # lambda
    assert len(zerocheck["polys"]) == max_log_row_count
    t.sumcheck(zerocheck, 4)
    t.observe([len(opened)])
    for name, prep, mainv, _ in opened:
        t.observe_var_exts(prep)
        t.observe_var_exts(mainv)

    # jagged verify_trusted_evaluations
    col_counts = [[c for _, c in rnd] for rnd in rc]
    n_prefix = sum(sum(c) for c in col_counts) + 1          # usize_prefix_sums.len()
    num_col_variables = (n_prefix - 1 - 1).bit_length() if n_prefix > 2 else 0
    jagged_start_op = len(t.ops)
    z_col = [t.sample_ext() for _ in range(num_col_variables)]
    t.sumcheck(jagged_sc, 2)
    t.observe_exts([jagged_eval_sc["claimed_sum"]])
    t.sumcheck(jagged_eval_sc, 2)
    # stacked: verify_untrusted_evaluation observes the claim; basefold observes the batch evaluations
    t.observe_exts([expected_eval])
    for be in batch_evals:
        t.observe_exts(be)
    t.check_witness(BATCH_GRINDING_BITS, batch_witness)
    total = sum(len(be) for be in batch_evals)
    for _ in range((total - 1).bit_length()):
        t.sample_ext()                                      # batching point
    t.observe([n_uni])
    for k in range(n_uni):
        t.observe_exts(uni[k])
        t.observe(fri_commits[k])
        t.sample_ext(expect=[int(v) for v in gold["betas"][k]])
    t.observe_exts([final_poly])
    t.check_witness(POW_BITS, pow_witness)
    qi = [int(v) for v in gold["query_indices"]]
    for q in range(NUM_QUERIES):
        t.sample_bits(n_uni + LOG_BLOWUP, expect=qi[q] if q < len(qi) else None)
    print("BaseFold transcript OK: betas, 16-bit PoW witness and the %d recovered query indices reproduce" % len(qi))

    ops = np.array(t.ops, dtype=np.int32)
    data = np.array(t.data, dtype=np.uint32)
    path = os.path.join(HERE, "kb_shrink_transcript.npz")
    log_stacking_height = n_uni
    stack_point = jagged_sc["point"][len(jagged_sc["point"]) - log_stacking_height:]
    np.savez_compressed(path, ops=ops, data=data, final_state=np.array(t.ch.state, dtype=np.uint32),
                        beta_seed_dim=np.int32(found), z_col_dim=np.int32(num_col_variables),
                        basefold_proof_q12=np.frombuffer(bytes(blob), dtype=np.uint8),
                        stack_point=np.array(stack_point, dtype=np.uint32),
                        expected_eval=np.array(expected_eval, dtype=np.uint32),
                        jagged_tail=np.frombuffer(bytes(jagged_tail), dtype=np.uint8),
                        shard_head=np.frombuffer(bytes(b[112:bf_start]), dtype=np.uint8),
                        gkr_range=np.array([gkr_start - 112, gkr_start - 112 + len(gkr_bytes)], dtype=np.int64),
                        shard_start_op=np.int32(shard_start_op),
                        gkr_start_op=np.int32(len(head[0])), zerocheck_start_op=np.int32(zerocheck_start_op),
                        jagged_start_op=np.int32(jagged_start_op),
                        jagged_z_row=np.array(zerocheck["point"], dtype=np.uint32),
                        jagged_claims0=np.array([e for _, prep, _, _ in opened for e in prep], dtype=np.uint32).reshape(-1, 4),
                        jagged_claims1=np.array([e for _, _, mainv, _ in opened for e in mainv], dtype=np.uint32).reshape(-1, 4))
    print("wrote", path, os.path.getsize(path), "bytes;", len(ops), "ops,", len(data), "words,",
          int((ops[:, 3] == 1).sum()), "pinned ops")


if __name__ == "__main__":
    main()
