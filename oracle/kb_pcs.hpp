// oracle/kb_pcs.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// CPU restatement (row-major layouts, exactly as the reference's CpuBackend) of the commit/open
// half of the core-shard hot path:
//   rs_encode                 /root/reference/slop/crates/basefold-prover/src/encoder.rs:L22-L38
//                             /root/reference/slop/crates/dft/src/p3.rs:L11-L49 (zero-pad, DFT, bit-reversed rows)
//   MerkleTree (commit/open/verify) /root/reference/slop/crates/merkle-tree/src/p3sync.rs:L40-L170,
//                             /root/reference/slop/crates/merkle-tree/src/tcs.rs:L102-L188
//   partial_lagrange          /root/reference/slop/crates/multilinear/src/lagrange.rs:L19-L45
//   eval_mle_at_point         /root/reference/slop/crates/multilinear/src/eval.rs:L9-L21
//   fold_mle                  /root/reference/slop/crates/multilinear/src/fold.rs:L12-L26
//   mle_fixed_at_zero         /root/reference/slop/crates/multilinear/src/restrict.rs:L75-L87
//   fold_even_odd             p3_fri::fold_even_odd (un-vendored, =0.4.3-succinct), pinned by the
//                             verifier's formula /root/reference/slop/crates/basefold/src/verifier.rs:L323-L388
//   FriCpuProver::{batch, commit_phase_round}  /root/reference/slop/crates/basefold-prover/src/fri.rs:L31-L129
//   BasefoldProver::{commit_mles, prove_trusted_mle_evaluations}
//                             /root/reference/slop/crates/basefold-prover/src/prover.rs:L78-L243
//   BasefoldVerifier::verify_mle_evaluations   /root/reference/slop/crates/basefold/src/verifier.rs:L122-L305
//   interleave_multilinears_with_fixed_rate    /root/reference/slop/crates/stacked/src/fixed_rate.rs:L6-L47
//   jagged commit wrapper     /root/reference/slop/crates/jagged/src/prover.rs:L106-L160
//   bincode proof layout      /root/reference/slop/crates/basefold/src/verifier.rs:L94-L116 (+ tcs.rs:L49-L91)
#pragma once
#include <algorithm>
#include <cassert>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kb_hash.hpp"
#include "kb_simd.hpp"

namespace orc {

// ------------------------------------------------------------------ Reed–Solomon encode
// in: [n][w] row-major coefficients; out: [n << log_blowup][w]; out[bitrev(k)] = sum_i in[i] w_N^{ki}
static inline void rs_encode(const F* in, int log_n, int w, int log_blowup, F* out) {
    const int log_N = log_n + log_blowup;
    const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
    memcpy(out, in, n * w * sizeof(F));
    memset((void*)(out + n * w), 0, (N - n) * w * sizeof(F));
    if (log_N == 0) return;
    std::vector<F> tw(N / 2);
    {
        F g = two_adic_generator(log_N), cur = F::one();
        for (size_t i = 0; i < N / 2; i++) { tw[i] = cur; cur *= g; }
    }
    // decimation-in-frequency: natural order in, bit-reversed order out
    const bool use_simd = simd::simd_available();
    for (int s = log_N; s >= 1; s--) {
        const size_t half = (size_t)1 << (s - 1), stride = N >> s;
#pragma omp parallel for schedule(static)
        for (size_t idx = 0; idx < N / 2; idx++) {
            size_t blk = idx / half, j = idx % half;
            F* a = out + (blk * 2 * half + j) * w;
            F* b = a + half * w;
            F t = tw[j * stride];
            if (use_simd && w >= 16) { simd::butterfly_row(a, b, t, w); continue; }
            for (int c = 0; c < w; c++) {
                F x = a[c], y = b[c];
                a[c] = x + y;
                b[c] = (x - y) * t;
            }
        }
    }
}

// ------------------------------------------------------------------ Merkle tensor commitment
struct TensorRef {
    const F* data;
    size_t height;
    int width;
};

struct MerkleTree {
    int log_height = 0;
    size_t total_width = 0;
    std::vector<std::vector<Digest>> layers;  // leaf-first; layers.back() = {root}
    Digest root, commit;
};

static inline MerkleTree merkle_commit(const std::vector<TensorRef>& ts) {
    assert(!ts.empty());
    MerkleTree mt;
    const size_t h = ts[0].height;
    for (auto& t : ts) { assert(t.height == h); mt.total_width += t.width; }
    assert((h & (h - 1)) == 0);
    mt.log_height = 0;
    while (((size_t)1 << mt.log_height) < h) mt.log_height++;
    std::vector<Digest> cur(h);
    const bool use_simd = simd::simd_available();
    if (use_simd && h >= 16 && h % 16 == 0) {                // sixteen rows per permutation (kb_simd.hpp); any other height: the scalar sponge
        std::vector<simd::RowSrc> src;
        for (auto& t : ts) src.push_back(simd::RowSrc{t.data, t.width});
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < h; i += 16) simd::hash_rows16(src.data(), src.size(), i, cur.data() + i);
    } else {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < h; i++) {
            Sponge sp;
            for (auto& t : ts)
                for (int c = 0; c < t.width; c++) sp.absorb(t.data[i * t.width + c]);
            cur[i] = sp.finish();
        }
    }
    mt.layers.push_back(cur);
    while (mt.layers.back().size() > 1) {
        const std::vector<Digest>& prev = mt.layers.back();
        std::vector<Digest> next(prev.size() / 2);
        if (use_simd && next.size() >= 16 && next.size() % 16 == 0) {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < next.size(); i += 16) simd::compress16(prev.data(), i, next.data());
        } else {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < next.size(); i++) next[i] = compress(prev[2 * i], prev[2 * i + 1]);
        }
        mt.layers.push_back(std::move(next));
    }
    mt.root = mt.layers.back()[0];
    F meta[2] = {F::from_canonical((uint32_t)mt.log_height), F::from_canonical((uint32_t)mt.total_width)};
    mt.commit = compress(mt.root, hash_slice(meta, 2));
    return mt;
}

struct TcsProof {
    Digest merkle_root;
    size_t log_tensor_height, width;
    std::vector<Digest> paths;  // [n_idx][log_height]
};

static inline TcsProof merkle_prove_openings(const MerkleTree& mt, const std::vector<size_t>& idx) {
    TcsProof p;
    p.merkle_root = mt.root;
    p.log_tensor_height = mt.log_height;
    p.width = mt.total_width;
    const size_t height = mt.layers.size() - 1;
    for (size_t i : idx)
        for (size_t k = 0; k < height; k++) p.paths.push_back(mt.layers[k][(i >> k) ^ 1]);
    return p;
}

// values: [n_idx][total_width] (all tensors' rows concatenated in message order)
static inline std::vector<F> compute_openings(const std::vector<TensorRef>& ts, const std::vector<size_t>& idx) {
    size_t tw = 0;
    for (auto& t : ts) tw += t.width;
    std::vector<F> out(idx.size() * tw);
    for (size_t q = 0; q < idx.size(); q++) {
        size_t off = 0;
        for (auto& t : ts) {
            memcpy(&out[q * tw + off], t.data + idx[q] * t.width, t.width * sizeof(F));
            off += t.width;
        }
    }
    return out;
}

enum class TcsError { Ok, RootMismatch, IncorrectShape, InconsistentCommitmentShape, IncorrectLogHeight, IncorrectWidth };

static inline TcsError merkle_verify(const Digest& commit, const std::vector<size_t>& idx, const F* opening,
                                     size_t opening_width, size_t expected_width, size_t expected_log_height,
                                     const TcsProof& proof) {
    if (proof.width != expected_width) return TcsError::IncorrectWidth;
    if (proof.log_tensor_height != expected_log_height) return TcsError::IncorrectLogHeight;
    if (proof.paths.size() != idx.size() * proof.log_tensor_height) return TcsError::IncorrectShape;
    if (opening_width != proof.width) return TcsError::IncorrectShape;
    for (size_t q = 0; q < idx.size(); q++) {
        Digest node = hash_slice(opening + q * opening_width, opening_width);
        size_t index = idx[q];
        for (size_t k = 0; k < proof.log_tensor_height; k++) {
            const Digest& sib = proof.paths[q * proof.log_tensor_height + k];
            node = (index & 1) == 0 ? compress(node, sib) : compress(sib, node);
            index >>= 1;
        }
        if (node != proof.merkle_root) return TcsError::RootMismatch;
        if (index != 0) return TcsError::IncorrectShape;
    }
    F meta[2] = {F::from_canonical((uint32_t)proof.log_tensor_height), F::from_canonical((uint32_t)proof.width)};
    if (compress(proof.merkle_root, hash_slice(meta, 2)) != commit) return TcsError::InconsistentCommitmentShape;
    return TcsError::Ok;
}

// ------------------------------------------------------------------ multilinear helpers
// eq(point, i), i big-endian: first coordinate = most significant bit.
static inline std::vector<E> partial_lagrange(const std::vector<E>& point) {
    std::vector<E> ev{E::one()};
    for (const E& x : point) {
        std::vector<E> nx(ev.size() * 2);
        for (size_t i = 0; i < ev.size(); i++) {
            E prod = ev[i] * x;
            nx[2 * i] = ev[i] - prod;
            nx[2 * i + 1] = prod;
        }
        ev.swap(nx);
    }
    return ev;
}

// base-field mle [n][w] row-major -> w extension evaluations
static inline std::vector<E> eval_mle_at_point(const F* mle, size_t n, int w, const std::vector<E>& point) {
    assert(((size_t)1 << point.size()) == n);
    std::vector<E> eq = partial_lagrange(point);
    std::vector<E> acc(w, E::zero());
#pragma omp parallel
    {
        std::vector<E> loc(w, E::zero());
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < n; i++)
            for (int c = 0; c < w; c++) loc[c] += eq[i] * mle[i * w + c];
#pragma omp critical
        for (int c = 0; c < w; c++) acc[c] += loc[c];
    }
    return acc;
}

static inline E eval_ext_mle_at_point(const std::vector<E>& mle, const std::vector<E>& point) {
    std::vector<E> eq = partial_lagrange(point);
    assert(eq.size() == mle.size());
    E acc = E::zero();
    for (size_t i = 0; i < mle.size(); i++) acc += eq[i] * mle[i];
    return acc;
}

static inline std::vector<E> fold_mle(const std::vector<E>& m, const E& beta) {
    std::vector<E> out(m.size() / 2);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < out.size(); i++) out[i] = m[2 * i] + beta * m[2 * i + 1];
    return out;
}

static inline E mle_fixed_at_zero(const std::vector<E>& m, const std::vector<E>& point) {
    std::vector<E> even(m.size() / 2);
    for (size_t i = 0; i < even.size(); i++) even[i] = m[2 * i];
    return eval_ext_mle_at_point(even, point);
}

// folded[i] = (1/2 + beta/(2 x_i)) cw[2i] + (1/2 - beta/(2 x_i)) cw[2i+1], x_i = w_N^{bitrev_{logN}(2i)}
// == interpolate (x_i, cw[2i]), (-x_i, cw[2i+1]) and evaluate at beta (verifier.rs:L364-L374).
static inline std::vector<E> fold_even_odd(const std::vector<E>& cw, const E& beta) {
    const size_t N = cw.size();
    int log_N = 0;
    while (((size_t)1 << log_N) < N) log_N++;
    const F g = two_adic_generator(log_N);
    const F half = finv(F::two());
    std::vector<E> out(N / 2);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N / 2; i++) {
        F x = fpow(g, reverse_bits_len((uint32_t)(2 * i), log_N));
        F inv2x = finv(x + x);
        const E &e0 = cw[2 * i], &e1 = cw[2 * i + 1];
        out[i] = (e0 + e1) * half + (beta * (e0 - e1)) * inv2x;
    }
    return out;
}

// ------------------------------------------------------------------ BaseFold
struct FriConfig {
    int log_blowup = 2, num_queries = 124, proof_of_work_bits = 16;
};
constexpr int BATCH_GRINDING_BITS = 5;

struct MleRef {  // base-field multilinear, [n][width] row-major
    const F* data;
    int log_n;
    int width;
};

struct BasefoldProverData {
    std::vector<std::vector<F>> codewords;  // per mle: [N][width]
    std::vector<int> widths;
    int log_N = 0;
    MerkleTree tree;
    std::vector<TensorRef> refs() const {
        std::vector<TensorRef> r;
        for (size_t i = 0; i < codewords.size(); i++)
            r.push_back({codewords[i].data(), (size_t)1 << log_N, widths[i]});
        return r;
    }
};

static inline std::shared_ptr<BasefoldProverData> commit_mles(const std::vector<MleRef>& mles, const FriConfig& cfg) {
    auto pd = std::make_shared<BasefoldProverData>();
    pd->log_N = mles[0].log_n + cfg.log_blowup;
    for (auto& m : mles) {
        assert(m.log_n == mles[0].log_n);
        std::vector<F> cw(((size_t)1 << pd->log_N) * m.width);
        rs_encode(m.data, m.log_n, m.width, cfg.log_blowup, cw.data());
        pd->codewords.push_back(std::move(cw));
        pd->widths.push_back(m.width);
    }
    pd->tree = merkle_commit(pd->refs());
    return pd;
}

struct OpeningAndProof {
    std::vector<F> values;  // [n_idx][width]
    size_t n_idx = 0, width = 0;
    TcsProof proof;
};

struct BasefoldProof {
    std::vector<std::array<E, 2>> univariate_messages;
    std::vector<Digest> fri_commitments;
    std::vector<OpeningAndProof> component_openings;
    std::vector<OpeningAndProof> query_phase_openings;
    E final_poly;
    F pow_witness, batch_grinding_witness;
};

static inline std::vector<E> sample_point(Challenger& ch, size_t n) {
    std::vector<E> p(n);
    for (auto& x : p) x = ch.sample_ext();
    return p;
}

static inline int log2_ceil(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}

// rounds[r] = mles of commitment round r; claims[r][m] = evaluations (one per column) of mle m
static inline BasefoldProof basefold_prove(std::vector<E> eval_point,
                                           const std::vector<std::vector<MleRef>>& rounds,
                                           const std::vector<std::vector<std::vector<E>>>& claims,
                                           const std::vector<std::shared_ptr<BasefoldProverData>>& pdata,
                                           const FriConfig& cfg, Challenger& ch) {
    BasefoldProof proof;
    std::vector<MleRef> mles;
    for (auto& r : rounds) for (auto& m : r) mles.push_back(m);
    std::vector<E> flat_claims;
    for (auto& r : claims) for (auto& m : r) for (auto& e : m) flat_claims.push_back(e);

    proof.batch_grinding_witness = ch.grind(BATCH_GRINDING_BITS);
    size_t total_len = 0;
    for (auto& m : mles) total_len += m.width;
    std::vector<E> coeffs = partial_lagrange(sample_point(ch, log2_ceil(total_len)));

    // FriCpuProver::batch
    const int nv = mles[0].log_n;
    const size_t n = (size_t)1 << nv;
    std::vector<E> cur_mle(n, E::zero());
    {
        size_t off = 0;
        for (auto& m : mles) {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) {
                E acc = E::zero();
                for (int c = 0; c < m.width; c++) acc += coeffs[off + c] * m.data[i * m.width + c];
                cur_mle[i] += acc;
            }
            off += m.width;
        }
    }
    E cur_claim = E::zero();
    for (size_t i = 0; i < flat_claims.size(); i++) cur_claim += flat_claims[i] * coeffs[i];
    // encode the batched mle as an [n][4] base matrix
    const size_t N0 = n << cfg.log_blowup;
    std::vector<E> cur_cw(N0);
    static_assert(sizeof(E) == 4 * sizeof(F), "E must be 4 packed F");
    rs_encode(reinterpret_cast<const F*>(cur_mle.data()), nv, 4, cfg.log_blowup, reinterpret_cast<F*>(cur_cw.data()));

    assert((size_t)nv == eval_point.size());
    ch.observe(F::from_canonical((uint32_t)eval_point.size()));
    std::vector<MerkleTree> trees;
    std::vector<std::vector<E>> leaves_per_round;
    const size_t dim = eval_point.size();
    for (size_t r = 0; r < dim; r++) {
        E last = eval_point.back();
        eval_point.pop_back();
        E zero_val = mle_fixed_at_zero(cur_mle, eval_point);
        E one_val = (cur_claim - zero_val) * einv(last) + zero_val;
        proof.univariate_messages.push_back({zero_val, one_val});
        ch.observe_ext(zero_val);
        ch.observe_ext(one_val);
        // commit_phase_round: leaves = codeword reshaped [N/2][8]
        TensorRef leaves{reinterpret_cast<const F*>(cur_cw.data()), cur_cw.size() / 2, 8};
        MerkleTree t = merkle_commit({leaves});
        ch.observe_digest(t.commit);
        E beta = ch.sample_ext();
        leaves_per_round.push_back(cur_cw);
        cur_cw = fold_even_odd(cur_cw, beta);
        cur_mle = fold_mle(cur_mle, beta);
        proof.fri_commitments.push_back(t.commit);
        trees.push_back(std::move(t));
        cur_claim = zero_val + beta * one_val;
    }
    proof.final_poly = cur_cw[0];
    ch.observe_ext(proof.final_poly);
    proof.pow_witness = ch.grind(cfg.proof_of_work_bits);
    std::vector<size_t> q(cfg.num_queries);
    for (auto& x : q) x = ch.sample_bits(nv + cfg.log_blowup);
    for (auto& pd : pdata) {
        OpeningAndProof o;
        auto refs = pd->refs();
        o.values = compute_openings(refs, q);
        o.n_idx = q.size();
        o.width = pd->tree.total_width;
        o.proof = merkle_prove_openings(pd->tree, q);
        proof.component_openings.push_back(std::move(o));
    }
    for (size_t r = 0; r < dim; r++) {
        for (auto& x : q) x >>= 1;
        TensorRef leaves{reinterpret_cast<const F*>(leaves_per_round[r].data()), leaves_per_round[r].size() / 2, 8};
        OpeningAndProof o;
        o.values = compute_openings({leaves}, q);
        o.n_idx = q.size();
        o.width = 8;
        o.proof = merkle_prove_openings(trees[r], q);
        proof.query_phase_openings.push_back(std::move(o));
    }
    return proof;
}

enum class BfError { Ok, SumcheckFriLengthMismatch, Tcs, Sumcheck, Pow, BatchPow, QueryValueMismatch,
                     QueryFinalPolyMismatch, SumcheckFinalPolyMismatch, IncorrectShape, TwoAdicityOverflow };

// claims[r] = all column evaluations of commitment round r, flattened in mle order
static inline BfError basefold_verify(const std::vector<Digest>& commitments, std::vector<E> point,
                                      const std::vector<std::vector<E>>& claims, const BasefoldProof& proof,
                                      const FriConfig& cfg, Challenger& ch) {
    if (!ch.check_witness(BATCH_GRINDING_BITS, proof.batch_grinding_witness)) return BfError::BatchPow;
    size_t total_len = 0;
    for (auto& c : claims) total_len += c.size();
    std::vector<E> coeffs = partial_lagrange(sample_point(ch, log2_ceil(total_len)));
    E eval_claim = E::zero();
    {
        size_t k = 0;
        for (auto& c : claims) for (auto& e : c) eval_claim += e * coeffs[k++];
    }
    if (claims.size() != commitments.size() || commitments.size() != proof.component_openings.size())
        return BfError::IncorrectShape;
    if (proof.fri_commitments.size() != proof.univariate_messages.size() ||
        proof.fri_commitments.size() != point.size() || proof.univariate_messages.empty())
        return BfError::SumcheckFriLengthMismatch;
    std::reverse(point.begin(), point.end());
    const size_t len = proof.fri_commitments.size();
    ch.observe(F::from_canonical((uint32_t)len));
    std::vector<E> betas;
    for (size_t i = 0; i < len; i++) {
        ch.observe_ext(proof.univariate_messages[i][0]);
        ch.observe_ext(proof.univariate_messages[i][1]);
        ch.observe_digest(proof.fri_commitments[i]);
        betas.push_back(ch.sample_ext());
    }
    auto& first = proof.univariate_messages[0];
    if (eval_claim != (E::one() - point[0]) * first[0] + point[0] * first[1]) return BfError::Sumcheck;
    E expected = first[0] + betas[0] * first[1];
    for (size_t i = 1; i < len; i++) {
        auto& poly = proof.univariate_messages[i];
        if (expected != (E::one() - point[i]) * poly[0] + point[i] * poly[1]) return BfError::Sumcheck;
        expected = poly[0] + betas[i] * poly[1];
    }
    ch.observe_ext(proof.final_poly);
    if (!ch.check_witness(cfg.proof_of_work_bits, proof.pow_witness)) return BfError::Pow;
    const size_t log_len = len;
    if ((int)(log_len + cfg.log_blowup) > KB_TWO_ADICITY) return BfError::TwoAdicityOverflow;
    std::vector<size_t> q(cfg.num_queries);
    for (auto& x : q) x = ch.sample_bits((int)log_len + cfg.log_blowup);

    std::vector<E> batch_evals(q.size(), E::zero());
    size_t batch_idx = 0;
    for (size_t r = 0; r < proof.component_openings.size(); r++) {
        auto& o = proof.component_openings[r];
        size_t total_columns = claims[r].size();
        if (o.n_idx != q.size() || o.width != total_columns || o.values.size() != o.n_idx * o.width)
            return BfError::IncorrectShape;
        for (size_t k = 0; k < q.size(); k++)
            for (size_t c = 0; c < total_columns; c++)
                batch_evals[k] += coeffs[batch_idx + c] * o.values[k * o.width + c];
        batch_idx += total_columns;
    }
    for (size_t r = 0; r < commitments.size(); r++) {
        auto& o = proof.component_openings[r];
        if (merkle_verify(commitments[r], q, o.values.data(), o.width, o.width, log_len + cfg.log_blowup, o.proof) !=
            TcsError::Ok)
            return BfError::Tcs;
    }
    // verify_queries
    const int log_max_height = (int)len + cfg.log_blowup;
    std::vector<E> folded = batch_evals;
    std::vector<size_t> idx = q;
    std::vector<F> xis(q.size());
    {
        F g = two_adic_generator(log_max_height);
        for (size_t k = 0; k < q.size(); k++) xis[k] = fpow(g, reverse_bits_len((uint32_t)q[k], log_max_height));
    }
    if (len != proof.query_phase_openings.size()) return BfError::IncorrectShape;
    const F minus_one = two_adic_generator(1);
    for (size_t r = 0; r < len; r++) {
        const int round_idx = log_max_height - 1 - (int)r;
        auto& o = proof.query_phase_openings[r];
        if (o.n_idx != idx.size() || o.width != 8 || o.values.size() != o.n_idx * 8) return BfError::IncorrectShape;
        for (size_t k = 0; k < idx.size(); k++) {
            size_t sib = idx[k] ^ 1, pair = idx[k] >> 1;
            E ev[2];
            memcpy(ev, &o.values[k * 8], sizeof ev);
            if (ev[idx[k] % 2] != folded[k]) return BfError::QueryValueMismatch;
            F xs[2] = {xis[k], xis[k]};
            xs[sib % 2] *= minus_one;
            folded[k] = ev[0] + ((betas[r] - xs[0]) * (ev[1] - ev[0])) * finv(xs[1] - xs[0]);
            idx[k] = pair;
            xis[k] *= xis[k];
        }
        if (merkle_verify(proof.fri_commitments[r], idx, o.values.data(), 8, 8, (size_t)round_idx, o.proof) !=
            TcsError::Ok)
            return BfError::Tcs;
    }
    for (auto& f : folded)
        if (f != proof.final_poly) return BfError::QueryFinalPolyMismatch;
    auto& lastp = proof.univariate_messages.back();
    if (proof.final_poly != lastp[0] + betas.back() * lastp[1]) return BfError::SumcheckFinalPolyMismatch;
    return BfError::Ok;
}

// ------------------------------------------------------------------ bincode (serde) of BasefoldProof
struct ByteWriter {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void u32(uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void f(F x) { u32(x.canonical()); }
    void e(const E& x) { for (int i = 0; i < 4; i++) f(x.c[i]); }
    void d(const Digest& x) { for (int i = 0; i < 8; i++) f(x.d[i]); }
};

static inline void write_opening(ByteWriter& w, const OpeningAndProof& o) {
    w.u64(o.values.size());
    for (auto& x : o.values) w.f(x);
    w.u64(2); w.u64(o.n_idx); w.u64(o.width);
    w.d(o.proof.merkle_root);
    w.u64(o.proof.log_tensor_height);
    w.u64(o.proof.width);
    w.u64(o.proof.paths.size());
    for (auto& x : o.proof.paths) w.d(x);
    w.u64(2); w.u64(o.n_idx); w.u64(o.proof.log_tensor_height);
}

static inline std::vector<uint8_t> serialize_proof(const BasefoldProof& p) {
    ByteWriter w;
    w.u64(p.univariate_messages.size());
    for (auto& m : p.univariate_messages) { w.e(m[0]); w.e(m[1]); }
    w.u64(p.fri_commitments.size());
    for (auto& c : p.fri_commitments) w.d(c);
    w.u64(p.component_openings.size());
    for (auto& o : p.component_openings) write_opening(w, o);
    w.u64(p.query_phase_openings.size());
    for (auto& o : p.query_phase_openings) write_opening(w, o);
    w.e(p.final_poly);
    w.f(p.pow_witness);
    w.f(p.batch_grinding_witness);
    return w.b;
}

struct ByteReader {
    const uint8_t* p;
    size_t n, o = 0;
    uint64_t u64() { need(8); uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[o + i] << (8 * i); o += 8; return v; }
    uint32_t u32() { need(4); uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[o + i] << (8 * i); o += 4; return v; }
    void need(size_t k) { if (o + k > n) throw std::runtime_error("proof blob truncated"); }
    F f() { uint32_t c = u32(); if (c >= KB_P) throw std::runtime_error("non-canonical felt"); return F::from_canonical(c); }
    E e() { E x; for (int i = 0; i < 4; i++) x.c[i] = f(); return x; }
    Digest d() { Digest x; for (int i = 0; i < 8; i++) x.d[i] = f(); return x; }
};

static inline OpeningAndProof read_opening(ByteReader& r) {
    OpeningAndProof o;
    size_t n = r.u64();
    if (n > r.n) throw std::runtime_error("bad length");
    o.values.resize(n);
    for (auto& x : o.values) x = r.f();
    if (r.u64() != 2) throw std::runtime_error("values must be 2-D");
    o.n_idx = r.u64(); o.width = r.u64();
    if (o.n_idx * o.width != n) throw std::runtime_error("bad values shape");
    o.proof.merkle_root = r.d();
    o.proof.log_tensor_height = r.u64();
    o.proof.width = r.u64();
    size_t np = r.u64();
    if (np > r.n) throw std::runtime_error("bad length");
    o.proof.paths.resize(np);
    for (auto& x : o.proof.paths) x = r.d();
    if (r.u64() != 2) throw std::runtime_error("paths must be 2-D");
    size_t a = r.u64(), b = r.u64();
    if (a * b != np) throw std::runtime_error("bad paths shape");
    return o;
}

static inline BasefoldProof read_basefold_proof(ByteReader& r) {
    const size_t len = r.n;
    BasefoldProof p;
    size_t n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    p.univariate_messages.resize(n);
    for (auto& m : p.univariate_messages) { m[0] = r.e(); m[1] = r.e(); }
    n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    p.fri_commitments.resize(n);
    for (auto& c : p.fri_commitments) c = r.d();
    n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    for (size_t i = 0; i < n; i++) p.component_openings.push_back(read_opening(r));
    n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    for (size_t i = 0; i < n; i++) p.query_phase_openings.push_back(read_opening(r));
    p.final_poly = r.e();
    p.pow_witness = r.f();
    p.batch_grinding_witness = r.f();
    return p;
}

static inline BasefoldProof deserialize_proof(const uint8_t* buf, size_t len) {
    ByteReader r{buf, len};
    BasefoldProof p = read_basefold_proof(r);
    if (r.o != len) throw std::runtime_error("trailing bytes");
    return p;
}

// ------------------------------------------------------------------ stacked interleave + jagged wrapper
// tables[k]: [rows_k][cols_k] row-major. Returns stacked batches, each [2^lsh][<=batch_size] row-major.
struct StackedBatch {
    std::vector<F> data;
    int width;
};
static inline std::vector<StackedBatch> interleave_fixed_rate(size_t batch_size,
                                                              const std::vector<TensorRef>& tables, int lsh) {
    // dense = column-major concatenation of all tables, cut into 2^lsh-long stacked columns,
    // zero-padded to a multiple of 2^lsh (at least one column); groups of batch_size columns.
    const size_t H = (size_t)1 << lsh;
    std::vector<F> dense;
    for (auto& t : tables)
        for (int c = 0; c < t.width; c++)
            for (size_t r = 0; r < t.height; r++) dense.push_back(t.data[r * t.width + c]);
    // The reference emits a full batch only while data.len() > needed, so a dense length that is an
    // exact multiple of batch_size*H leaves the last full batch in the overflow buffer, which is then
    // emitted as the final (full-width) batch: identical to plain chunking. An empty input yields one
    // zero-width batch (overflow_batch_size == 0), mirrored below.
    size_t padded = ((dense.size() + H - 1) / H) * H;
    dense.resize(padded, F::zero());
    const size_t ncols = padded / H;
    std::vector<StackedBatch> out;
    if (ncols == 0) out.push_back(StackedBatch{{}, 0});
    for (size_t c0 = 0; c0 < ncols; c0 += batch_size) {
        size_t w = std::min(batch_size, ncols - c0);
        StackedBatch b;
        b.width = (int)w;
        b.data.resize(H * w);
        for (size_t c = 0; c < w; c++)
            for (size_t r = 0; r < H; r++) b.data[r * w + c] = dense[(c0 + c) * H + r];
        out.push_back(std::move(b));
    }
    return out;
}

// final = compress(commit, hash([len, rows.., cols..])) with the two padding tables appended
static inline Digest jagged_commit_wrap(const Digest& commit, std::vector<size_t> rows, std::vector<size_t> cols,
                                        size_t num_added_vals, int max_log_row_count) {
    const size_t M = (size_t)1 << max_log_row_count;
    size_t num_added_cols = std::max<size_t>((num_added_vals + M - 1) / M, 1);
    rows.push_back(M);
    rows.push_back(num_added_vals - (num_added_cols - 1) * M);
    cols.push_back(num_added_cols - 1);
    cols.push_back(1);
    std::vector<F> in;
    in.push_back(F::from_canonical((uint32_t)rows.size()));
    for (size_t x : rows) in.push_back(F::from_canonical((uint32_t)x));
    for (size_t x : cols) in.push_back(F::from_canonical((uint32_t)x));
    return compress(commit, hash_slice(in.data(), in.size()));
}

}  // namespace orc
