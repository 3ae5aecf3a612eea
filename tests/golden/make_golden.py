#!/usr/bin/env python3
"""Extract a small golden fixture from the reference's one real proof artifact.

Source (read-only, only in the build container):
  /root/reference/sp1-gpu/crates/perf/recursion_records/shrink_input.bin
  = bincode(SP1ShapedWitnessValues{vks_and_proofs:[(MachineVerifyingKey, ShardProof<KoalaBear..>)],..})
  written by /root/reference/crates/prover/src/worker/prover/recursion.rs:L1099-L1111.

bincode layout facts used (all confirmed by the parse itself — every length/shape field is
checked): Vec = u64 LE length + items; field element = canonical u32 LE; Tensor = {storage: Vec,
dimensions: Vec<usize>} (/root/reference/slop/crates/tensor/src/inner.rs:L670-L677,
dimensions.rs:L159-L163); BasefoldProof field order as in
/root/reference/slop/crates/basefold/src/verifier.rs:L94-L116; MerkleTreeOpeningAndProof /
MerkleTreeTcsProof as in /root/reference/slop/crates/merkle-tree/src/tcs.rs:L49-L91;
JaggedPcsProof as in /root/reference/slop/crates/jagged/src/verifier.rs:L17-L26.

What goes into tests/golden/kb_shrink_basefold.npz (first NQ of the 124 queries only, to stay small):
  * both component openings (widths 34 and 52, log height 22) with Merkle paths + roots,
  * all 20 fold-round openings (width 8, log heights 21..2) with Merkle paths + roots,
  * the recovered query indices (the proof does not store them: they come from the Fiat-Shamir
    transcript; they are recovered here by walking each path with both left/right orders),
  * univariate messages, fri commitments, final_poly, the two PoW witnesses,
  * merkle_tree_commitments, row/column counts, vk.preprocessed_commit, proof.main_commitment,
  * the three PartialSumcheckProofs of the shard proof (zerocheck: 21 rounds x 5 coefficients; jagged
    sumcheck; jagged eval) — used to pin the univariate-message conventions of the sumcheck driver.

While extracting, the script *verifies* with oracle/kb_py.py (pure Python) that
  - every kept leaf hashes up its path to the stored root,
  - compress(root, hash([log_h, width])) equals the stored merkle_tree_commitments / fri_commitments,
  - compress(commit, hash([n, rows.., cols..])) equals vk.preprocessed_commit / main_commitment,
  - one beta per round solved from query 0 folds every other kept query onto the next round's
    opened value, and the last round onto final_poly = uni[-1][0] + beta*uni[-1][1].
So a successful run pins Poseidon2/sponge/compress/commit-chain/fold against real reference output.
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import kb_py as kb  # noqa: E402

SRC = "/root/reference/sp1-gpu/crates/perf/recursion_records/shrink_input.bin"
NQ = 12


class Reader:
    def __init__(self, buf, off):
        self.b, self.o = buf, off

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def u32s(self, k):
        v = struct.unpack_from("<%dI" % k, self.b, self.o)
        self.o += 4 * k
        return list(v)

    def opening(self):
        n = self.u64()
        vals = self.u32s(n)
        dims = [self.u64() for _ in range(self.u64())]
        assert len(dims) == 2 and dims[0] * dims[1] == n
        root = self.u32s(8)
        log_h, width = self.u64(), self.u64()
        npath = self.u64()
        paths = self.u32s(npath * 8)
        pd = [self.u64() for _ in range(self.u64())]
        assert pd == [dims[0], log_h] and npath == dims[0] * log_h and width == dims[1]
        vals = np.array(vals, dtype=np.uint32).reshape(dims)
        paths = np.array(paths, dtype=np.uint32).reshape(dims[0], log_h, 8)
        return dict(values=vals, root=root, log_h=log_h, width=width, paths=paths)

    def sumcheck(self):
        """PartialSumcheckProof<EF> (/root/reference/slop/crates/sumcheck/src/proof.rs:L10-L14)."""
        polys = []
        for _ in range(self.u64()):
            polys.append(self.u32s(4 * self.u64()))
        claimed = self.u32s(4)
        point = self.u32s(4 * self.u64())
        ev = self.u32s(4)
        assert len({len(p) for p in polys}) == 1 and len(point) == 4 * len(polys)
        return dict(polys=np.array(polys, dtype=np.uint32).reshape(len(polys), -1, 4),
                    claimed_sum=np.array(claimed, dtype=np.uint32), point=np.array(point, dtype=np.uint32).reshape(-1, 4),
                    eval=np.array(ev, dtype=np.uint32))


def find_zerocheck_proof(b, limit):
    """ShardProof.zerocheck_proof sits before the opened values: max_log_row_count (= 21 here) polys of 5
    coefficients each (degree 4)."""
    pat = struct.pack("<Q", 21) + struct.pack("<Q", 5)
    o = b.find(pat)
    while 0 <= o < limit:
        if struct.unpack_from("<Q", b, o + 8 + 88)[0] == 5 and struct.unpack_from("<Q", b, o + 8 + 88 * 20)[0] == 5:
            return o
        o = b.find(pat, o + 1)
    raise RuntimeError("zerocheck proof not found")


def find_basefold_start(b):
    pat = struct.pack("<Q", 20)
    o = b.find(pat)
    while o >= 0:
        if b[o + 648:o + 656] == pat and struct.unpack_from("<Q", b, o + 656 + 640)[0] == 2:
            return o
        o = b.find(pat, o + 1)
    raise RuntimeError("BasefoldProof not found")


def main():
    b = open(SRC, "rb").read()
    r = Reader(b, find_basefold_start(b))
    n_uni = r.u64()
    uni = np.array(r.u32s(n_uni * 8), dtype=np.uint32).reshape(n_uni, 2, 4)
    n_c = r.u64()
    fri_commitments = np.array(r.u32s(n_c * 8), dtype=np.uint32).reshape(n_c, 8)
    comps = [r.opening() for _ in range(r.u64())]
    rounds = [r.opening() for _ in range(r.u64())]
    final_poly = r.u32s(4)
    pow_witness, batch_witness = r.u32s(1)[0], r.u32s(1)[0]
    assert n_uni == n_c == len(rounds) == 20 and len(comps) == 2
    # StackedBasefoldProof.batch_evaluations, then the rest of JaggedPcsProof
    batch_evals = []
    for _ in range(r.u64()):
        n = r.u64()
        batch_evals.append(np.array(r.u32s(4 * n), dtype=np.uint32).reshape(n, 4))
        assert [r.u64() for _ in range(r.u64())] == [n]
    jagged_sumcheck = r.sumcheck()
    jagged_eval = r.sumcheck()
    zc = Reader(b, find_zerocheck_proof(b, r.o)).sumcheck()
    assert zc["polys"].shape == (21, 5, 4)
    rc = []
    for _ in range(r.u64()):
        rc.append([(r.u64(), r.u64()) for _ in range(r.u64())])
    mt_commits = [r.u32s(8) for _ in range(r.u64())]
    r.u32s(4)
    max_log_row_count, log_m = r.u64(), r.u64()
    assert len(rc) == len(mt_commits) == 2 and max_log_row_count == 21
    vk_pre_commit = list(struct.unpack_from("<8I", b, 76))
    main_commit = list(struct.unpack_from("<8I", b, 868))

    # ---- commitment chain (p3sync.rs:L136-L142, jagged/src/prover.rs:L141-L149) ----------------
    for k, (comp, final) in enumerate(zip(comps, (vk_pre_commit, main_commit))):
        c1 = kb.compress(comp["root"], kb.hash_felts([comp["log_h"], comp["width"]]))
        assert c1 == mt_commits[k], "tensor commitment mismatch"
        rows = [x for x, _ in rc[k]]
        cols = [y for _, y in rc[k]]
        c2 = kb.compress(c1, kb.hash_felts([len(rows)] + rows + cols))
        assert c2 == final, "jagged commitment mismatch"
    for k, rd in enumerate(rounds):
        c1 = kb.compress(rd["root"], kb.hash_felts([rd["log_h"], rd["width"]]))
        assert c1 == list(fri_commitments[k]), "fri commitment mismatch"
    print("commitment chains OK")

    # ---- recover query indices bottom-up, verifying every kept path ------------------------------
    def ok(idx, op, q):
        root, rest = kb.merkle_root_from_path(idx, [int(v) for v in op["values"][q]],
                                              [[int(v) for v in s] for s in op["paths"][q]])
        return rest == 0 and root == op["root"]

    indices = []
    for q in range(NQ):
        cand = [i for i in range(4) if ok(i, rounds[-1], q)]
        assert len(cand) == 1
        idx = cand[0]
        for rd in reversed(rounds[:-1]):
            nxt = [2 * idx + bit for bit in (0, 1) if ok(2 * idx + bit, rd, q)]
            assert len(nxt) == 1
            idx = nxt[0]
        full = [2 * idx + bit for bit in (0, 1) if all(ok(2 * idx + bit, c, q) for c in comps)]
        assert len(full) == 1
        indices.append(full[0])
        print("query", q, "index", full[0])

    # ---- fold consistency: solve beta from query 0, check the others -----------------------------
    log_max = 22
    betas = []
    for k, rd in enumerate(rounds):
        lh = log_max - k                      # codeword length 2^lh before this fold
        g = kb.two_adic_generator(lh)

        def pair(q):
            i = indices[q] >> k               # index into the round-k codeword
            vals = [int(v) for v in rd["values"][q]]
            e0, e1 = vals[:4], vals[4:]
            x0 = pow(g, kb.reverse_bits_len((i >> 1) << 1, lh), kb.P)
            return i, e0, e1, x0

        def target(q):
            if k + 1 < len(rounds):
                i_next = indices[q] >> (k + 1)
                vals = [int(v) for v in rounds[k + 1]["values"][q]]
                return vals[4 * (i_next & 1):4 * (i_next & 1) + 4]
            return final_poly

        i, e0, e1, x0 = pair(0)
        x1 = (kb.P - x0) % kb.P
        # beta = x0 + (target - e0) * (x1 - x0) / (e1 - e0)
        num = kb.ext_scale(kb.ext_sub(target(0), e0), (x1 - x0) % kb.P)
        beta = kb.ext_add(kb.ext_from_base(x0), kb.ext_mul(num, kb.ext_inv(kb.ext_sub(e1, e0))))
        for q in range(1, NQ):
            _, f0, f1, y0 = pair(q)
            assert kb.fold_query(f0, f1, beta, y0) == target(q), "fold mismatch"
        betas.append(beta)
    last = [int(v) for v in uni[-1].reshape(-1)]
    assert kb.ext_add(last[:4], kb.ext_mul(betas[-1], last[4:])) == final_poly
    print("fold rounds OK; final_poly consistent with last univariate message")

    out = dict(
        query_indices=np.array(indices, dtype=np.uint32),
        betas=np.array(betas, dtype=np.uint32),
        uni=uni, fri_commitments=fri_commitments,
        final_poly=np.array(final_poly, dtype=np.uint32),
        pow_witness=np.uint32(pow_witness), batch_witness=np.uint32(batch_witness),
        mt_commits=np.array(mt_commits, dtype=np.uint32),
        row_counts=np.array([[x for x, _ in t] for t in rc], dtype=np.uint32),
        col_counts=np.array([[y for _, y in t] for t in rc], dtype=np.uint32),
        vk_preprocessed_commit=np.array(vk_pre_commit, dtype=np.uint32),
        main_commitment=np.array(main_commit, dtype=np.uint32),
        batch_evals0=batch_evals[0], batch_evals1=batch_evals[1],
    )
    for name, sc in (("zerocheck", zc), ("jagged_sumcheck", jagged_sumcheck), ("jagged_eval", jagged_eval)):
        for key, val in sc.items():
            out["%s_%s" % (name, key)] = val
    for k, c in enumerate(comps):
        out["comp%d_values" % k] = c["values"][:NQ]
        out["comp%d_paths" % k] = c["paths"][:NQ]
        out["comp%d_root" % k] = np.array(c["root"], dtype=np.uint32)
    for k, rd in enumerate(rounds):
        out["round%02d_values" % k] = rd["values"][:NQ]
        out["round%02d_paths" % k] = rd["paths"][:NQ]
        out["round%02d_root" % k] = np.array(rd["root"], dtype=np.uint32)
    path = os.path.join(HERE, "kb_shrink_basefold.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
