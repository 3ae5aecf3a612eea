// oracle/kb_jagged.hpp — TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
//
// Jagged PCS evaluation proof: everything between the zerocheck point and the
// BaseFold opening. Restates, in the reference's own order of operations:
//   JaggedProver::prove_trusted_evaluations          /root/reference/slop/crates/jagged/src/prover.rs:L162-L328
//   jagged_sumcheck_poly / HadamardProduct           /root/reference/slop/crates/jagged/src/sumcheck.rs:L13-L39,
//                                                    hadamard.rs:L52-L146
//   JaggedLittlePolynomialProverParams               /root/reference/slop/crates/jagged/src/poly.rs:L238-L318
//   BranchingProgram::eval / transition_function     /root/reference/slop/crates/jagged/src/poly.rs:L120-L160,L385-L460
//   full_jagged_little_polynomial_evaluation         /root/reference/slop/crates/jagged/src/poly.rs:L183-L236
//   JaggedEvalSumcheckProver (CPU impl)              /root/reference/slop/crates/jagged/src/jagged_eval/sumcheck_eval.rs:L185-L243,
//                                                    sumcheck_poly.rs:L77-L170, sumcheck_sum_as_poly.rs:L60-L247,
//                                                    eval_sumcheck_prover.rs:L16-L79
//   reduce_sumcheck_to_evaluation                    /root/reference/slop/crates/sumcheck/src/prover.rs:L13-L96
//   partially_verify_sumcheck_proof                  /root/reference/slop/crates/sumcheck/src/verifier.rs:L21-L95
//   StackedPcsProver::prove_trusted_evaluation       /root/reference/slop/crates/stacked/src/prover.rs:L107-L152
//   BasefoldProver::prove_untrusted_evaluations      /root/reference/slop/crates/basefold-prover/src/prover.rs:L245-L270
//   JaggedPcsVerifier::verify_trusted_evaluations    /root/reference/slop/crates/jagged/src/verifier.rs:L109-L383
//   JaggedEvalSumcheckConfig::jagged_evaluation      /root/reference/slop/crates/jagged/src/jagged_eval/sumcheck_eval.rs:L47-L170
//   StackedPcsVerifier::verify_trusted_evaluation    /root/reference/slop/crates/stacked/src/verifier.rs:L39-L99
// Pinned by reference data: tests/test_oracle_golden.py runs jagged_verify on the reference's REAL
// JaggedPcsProof (tests/golden/kb_shrink_transcript.npz: its own bytes, BaseFold part restricted to 12
// queries) from the replayed transcript state — branching program, jagged-eval closing check, both
// sumchecks, claim insertion/padding, stacked interpolation and the BaseFold opening all accept.
#pragma once
#include <numeric>

#include "kb_pcs.hpp"
#include "kb_zerocheck.hpp"

namespace orc {

// ------------------------------------------------------------------ PartialSumcheckProof<EF>
struct SumcheckProof {
    std::vector<UniPoly> polys;
    E claimed_sum;
    std::vector<E> point;
    E eval;
};

static inline void write_sumcheck(ByteWriter& w, const SumcheckProof& p) {
    w.u64(p.polys.size());
    for (auto& u : p.polys) { w.u64(u.size()); for (auto& c : u) w.e(c); }
    w.e(p.claimed_sum);
    w.u64(p.point.size());
    for (auto& x : p.point) w.e(x);
    w.e(p.eval);
}

static inline SumcheckProof read_sumcheck(ByteReader& r) {
    SumcheckProof p;
    size_t n = r.u64();
    if (n > r.n) throw std::runtime_error("bad length");
    p.polys.resize(n);
    for (auto& u : p.polys) {
        size_t k = r.u64();
        if (k > r.n) throw std::runtime_error("bad length");
        u.resize(k);
        for (auto& c : u) c = r.e();
    }
    p.claimed_sum = r.e();
    n = r.u64();
    if (n > r.n) throw std::runtime_error("bad length");
    p.point.resize(n);
    for (auto& x : p.point) x = r.e();
    p.eval = r.e();
    return p;
}

// verifier.rs:L21-L95. Returns 0 or a positive error code.
static inline int partially_verify_sumcheck(const SumcheckProof& proof, Challenger& ch, size_t expected_vars,
                                            size_t expected_degree) {
    const size_t nv = proof.polys.size();
    if (nv != proof.point.size() || nv != expected_vars || expected_vars == 0) return 1;
    const UniPoly& first = proof.polys[0];
    if (uni_eval_one_plus_eval_zero(first) != proof.claimed_sum) return 2;
    if (first.size() != expected_degree + 1) return 1;
    for (auto& c : first) ch.observe_ext(c);
    std::vector<E> alphas;   // in sampling order; proof.point = reversed
    const UniPoly* prev = &first;
    for (size_t k = 1; k < nv; k++) {
        const UniPoly& poly = proof.polys[k];
        if (poly.size() != expected_degree + 1) return 1;
        E alpha = ch.sample_ext();
        alphas.push_back(alpha);
        if (uni_eval(*prev, alpha) != uni_eval_one_plus_eval_zero(poly)) return 3;
        for (auto& c : poly) ch.observe_ext(c);
        prev = &poly;
    }
    E alpha = ch.sample_ext();
    alphas.push_back(alpha);
    for (size_t k = 0; k < nv; k++)
        if (alphas[nv - 1 - k] != proof.point[k]) return 1;
    if (uni_eval(*prev, alpha) != proof.eval) return 4;
    return 0;
}

// ------------------------------------------------------------------ jagged little polynomial
static inline int log2_ceil_usize(size_t x) { return log2_ceil(x); }

struct JaggedParams {                       // JaggedLittlePolynomialProverParams
    std::vector<size_t> prefix;             // col_prefix_sums_usize: #columns + 1 entries
    int max_log_row_count = 0;
    static JaggedParams from_column_heights(const std::vector<size_t>& heights, int mlrc) {
        JaggedParams p;
        p.max_log_row_count = mlrc;
        size_t s = 0;
        for (size_t h : heights) { p.prefix.push_back(s); s += h; }
        p.prefix.push_back(s);
        return p;
    }
    int log_m() const { return log2_ceil_usize(prefix.back()); }
};

static inline std::vector<E> last_k(const std::vector<E>& p, size_t k) { return std::vector<E>(p.end() - k, p.end()); }

// poly.rs:L258-L318: guts of J(x) = eq(z_col, col(x)) eq(z_row, row(x)) over the 2^log_m dense indices,
// zero past the last prefix sum.
static inline std::vector<E> partial_jagged_table(const JaggedParams& pp, const std::vector<E>& z_row,
                                                  const std::vector<E>& z_col) {
    const size_t total = (size_t)1 << pp.log_m();
    const std::vector<E> col_eq = partial_lagrange(last_k(z_col, log2_ceil_usize(pp.prefix.size() - 1)));
    const std::vector<E> row_eq = partial_lagrange(last_k(z_row, pp.max_log_row_count));
    std::vector<E> out(total, E::zero());
    const size_t ncols = pp.prefix.size() - 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < ncols; c++)
        for (size_t i = pp.prefix[c]; i < pp.prefix[c + 1]; i++) out[i] = col_eq[c] * row_eq[i - pp.prefix[c]];
    return out;
}

// Branching program of HR18 for "index == prefix + row and index < next_prefix" (poly.rs:L120-L160).
// memory state index = carry + 2 * comparison_so_far; bit state index = row<<3 | index<<2 | curr<<1 | next.
struct BranchingProgram {
    std::vector<E> z_row, z_index;
    size_t num_vars;
    int trans[4][16];   // -1 = fail
    BranchingProgram(std::vector<E> zr, std::vector<E> zi) : z_row(std::move(zr)), z_index(std::move(zi)) {
        num_vars = std::max(z_row.size(), z_index.size());
        for (int m = 0; m < 4; m++)
            for (int b = 0; b < 16; b++) {
                const int carry = m & 1, cmp = m >> 1;
                const int row = (b >> 3) & 1, idx = (b >> 2) & 1, cur = (b >> 1) & 1, nxt = b & 1;
                const int new_cmp = idx == nxt ? cmp : nxt;
                const int s = row + carry + cur;
                trans[m][b] = (idx != (s & 1)) ? -1 : ((s >> 1) + 2 * new_cmp);
            }
    }
    static E ith_least_significant(const std::vector<E>& p, size_t i) { return p.size() <= i ? E::zero() : p[p.size() - i - 1]; }
    E eval(const std::vector<E>& prefix_sum, const std::vector<E>& next_prefix_sum) const {
        E res[4] = {E::zero(), E::zero(), E::one(), E::zero()};   // success = {carry 0, comparison 1}
        for (size_t layer = num_vars + 1; layer-- > 0;) {
            const std::vector<E> eq = partial_lagrange({ith_least_significant(z_row, layer), ith_least_significant(z_index, layer),
                                                        ith_least_significant(prefix_sum, layer),
                                                        ith_least_significant(next_prefix_sum, layer)});
            E nres[4];
            for (int m = 0; m < 4; m++) {
                E accum[4] = {E::zero(), E::zero(), E::zero(), E::zero()};
                for (int b = 0; b < 16; b++)
                    if (trans[m][b] >= 0) accum[trans[m][b]] += eq[b];
                E acc = E::zero();
                for (int k = 0; k < 4; k++) acc += accum[k] * res[k];
                nres[m] = acc;
            }
            for (int m = 0; m < 4; m++) res[m] = nres[m];
        }
        return res[0];
    }
};

static inline std::vector<E> point_from_usize(size_t x, size_t dim) {
    std::vector<E> p(dim);
    for (size_t i = 0; i < dim; i++) p[i] = ((x >> (dim - 1 - i)) & 1) ? E::one() : E::zero();
    return p;
}

// poly.rs:L183-L236
static inline E full_jagged_little_polynomial_evaluation(const std::vector<size_t>& prefix, const std::vector<E>& z_row,
                                                         const std::vector<E>& z_col, const std::vector<E>& z_index) {
    const int log_m = log2_ceil_usize(prefix.back());
    const std::vector<E> col_eq = partial_lagrange(z_col);
    const BranchingProgram bp(z_row, z_index);
    const size_t ncols = prefix.size() - 1;
    std::vector<E> terms(ncols);
#pragma omp parallel for schedule(dynamic, 8)
    for (size_t c = 0; c < ncols; c++)
        terms[c] = col_eq[c] * bp.eval(point_from_usize(prefix[c], log_m + 1), point_from_usize(prefix[c + 1], log_m + 1));
    E acc = E::zero();
    for (auto& t : terms) acc += t;
    return acc;
}

// ------------------------------------------------------------------ Hadamard-product sumcheck (degree 2)
// hadamard.rs:L100-L146: y(0) = sum of even-index products, y(1) = claim - y(0), y(1/2) = sum of
// (j0 + j1)(q0 + q1) / 4; interpolate on [0, 1, 1/2].
template <class K>
static inline UniPoly hadamard_sum_as_poly(const std::vector<K>& q, const std::vector<E>& j, const E& claim) {
    const size_t half = q.size() / 2;
    E e0 = E::zero(), eh = E::zero();
#pragma omp parallel
    {
        E l0 = E::zero(), lh = E::zero();
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < half; i++) {
            l0 += j[2 * i] * q[2 * i];
            lh += (j[2 * i] + j[2 * i + 1]) * (q[2 * i] + q[2 * i + 1]);
        }
#pragma omp critical
        { e0 += l0; eh += lh; }
    }
    const E e1 = claim - e0;
    const E two = E::from_base(F::two()), four = E::from_base(F::from_canonical(4));
    return interpolate_univariate({E::zero(), E::one(), einv(two)}, {e0, e1, eh * einv(four)});
}

template <class K>
static inline std::vector<E> fix_last_variable(const std::vector<K>& v, const E& alpha) {
    std::vector<E> out(v.size() / 2);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < out.size(); i++) out[i] = E::zero() + v[2 * i] + alpha * (E::zero() + v[2 * i + 1] - v[2 * i]);
    return out;
}

// reduce_sumcheck_to_evaluation for ONE HadamardProduct, t = 1, lambda = 1. Returns the proof and the
// two component evaluations [q(point), J(point)].
static inline SumcheckProof hadamard_sumcheck(const std::vector<F>& q, const std::vector<E>& j, const E& claim, Challenger& ch,
                                              E* q_eval, E* j_eval) {
    SumcheckProof proof;
    proof.claimed_sum = claim;
    size_t nv = 0;
    while (((size_t)1 << nv) < q.size()) nv++;
    std::vector<E> alphas;
    UniPoly uni = hadamard_sum_as_poly(q, j, claim);
    for (auto& c : uni) ch.observe_ext(c);
    proof.polys.push_back(uni);
    E alpha = ch.sample_ext();
    alphas.push_back(alpha);
    std::vector<E> qe = fix_last_variable(q, alpha), je = fix_last_variable(j, alpha);
    for (size_t r = 1; r < nv; r++) {
        const E round_claim = uni_eval(uni, alpha);
        uni = hadamard_sum_as_poly(qe, je, round_claim);
        for (auto& c : uni) ch.observe_ext(c);
        proof.polys.push_back(uni);
        alpha = ch.sample_ext();
        alphas.push_back(alpha);
        qe = fix_last_variable(qe, alpha);
        je = fix_last_variable(je, alpha);
    }
    proof.eval = uni_eval(uni, alpha);
    proof.point.assign(alphas.rbegin(), alphas.rend());
    *q_eval = qe[0];
    *j_eval = je[0];
    return proof;
}

// ------------------------------------------------------------------ jagged-eval sumcheck (prover)
// sumcheck_poly.rs:L77-L170 + sumcheck_sum_as_poly.rs + eval_sumcheck_prover.rs, CPU implementation.
// Variables: the 2 (log_m + 1) bits of (t_c || t_{c+1}); the summand is
//   sum_c eq(z_col, c) * eq(merged_c, x) * BP(z_row, z_trace; x_left, x_right).
static inline SumcheckProof jagged_eval_prove(const JaggedParams& pp, const std::vector<E>& z_row, const std::vector<E>& z_col,
                                              const std::vector<E>& z_trace, Challenger& ch) {
    const int log_m = pp.log_m();
    const size_t dim = 2 * (size_t)(log_m + 1);
    // merged prefix sums as bit strings, condensed over runs of equal (t_c, t_{c+1}) (empty tables)
    std::vector<std::vector<uint8_t>> merged;
    std::vector<E> z_col_eq_vals;
    {
        const std::vector<E> col_eq = partial_lagrange(z_col);
        const size_t ncols = pp.prefix.size() - 1;
        for (size_t c = 0; c < ncols; c++) {
            std::vector<uint8_t> bits(dim);
            for (int i = 0; i <= log_m; i++) {
                bits[i] = (pp.prefix[c] >> (log_m - i)) & 1;
                bits[log_m + 1 + i] = (pp.prefix[c + 1] >> (log_m - i)) & 1;
            }
            if (!merged.empty() && merged.back() == bits) z_col_eq_vals.back() += col_eq[c];
            else { merged.push_back(bits); z_col_eq_vals.push_back(col_eq[c]); }
        }
    }
    const size_t n = merged.size();
    const BranchingProgram bp(z_row, z_trace);
    const E half = einv(E::from_base(F::two()));
    const E expected_sum = full_jagged_little_polynomial_evaluation(pp.prefix, z_row, z_col, z_trace);
    ch.observe_ext(expected_sum);

    SumcheckProof proof;
    proof.claimed_sum = expected_sum;
    std::vector<E> inter(n, E::one());   // intermediate_eq_full_evals
    std::vector<E> rhos;                 // newest first (Point::add_dimension inserts at the front)
    E claim = expected_sum;
    UniPoly uni;
    for (size_t round = 0; round < dim; round++) {
        E y0 = E::zero(), yh = E::zero();
#pragma omp parallel
        {
            E l0 = E::zero(), lh = E::zero();
#pragma omp for schedule(dynamic, 4) nowait
            for (size_t k = 0; k < n; k++) {
                const size_t split = dim - round - 1;
                for (int which = 0; which < 2; which++) {
                    const E lambda = which ? half : E::zero();
                    const E eq_val = which ? half : (merged[k][split] ? E::zero() : E::one());
                    std::vector<E> h(dim);
                    for (size_t i = 0; i < split; i++) h[i] = merged[k][i] ? E::one() : E::zero();
                    h[split] = lambda;
                    for (size_t i = 0; i < rhos.size(); i++) h[split + 1 + i] = rhos[i];
                    const std::vector<E> left(h.begin(), h.begin() + dim / 2), right(h.begin() + dim / 2, h.end());
                    const E v = z_col_eq_vals[k] * bp.eval(left, right) * (inter[k] * eq_val);
                    if (which) lh += v; else l0 += v;
                }
            }
#pragma omp critical
            { y0 += l0; yh += lh; }
        }
        const E y1 = claim - y0;
        uni = interpolate_univariate({E::zero(), half, E::one()}, {y0, yh, y1});
        for (auto& c : uni) ch.observe_ext(c);
        proof.polys.push_back(uni);
        const E alpha = ch.sample_ext();
        rhos.insert(rhos.begin(), alpha);
        claim = uni_eval(uni, alpha);
        // fix_last_variable: fold the newly fixed coordinate into the eq accumulators
        for (size_t k = 0; k < n; k++) {
            const E x = merged[k][dim - 1 - round] ? E::one() : E::zero();
            inter[k] = inter[k] * (alpha * x + (E::one() - alpha) * (E::one() - x));
        }
    }
    proof.point = rhos;
    proof.eval = uni_eval(uni, rhos.front());
    return proof;
}

// jagged_eval/sumcheck_eval.rs:L47-L170 (verifier). Returns 0 and the claimed J(z_trace) in *out.
static inline int jagged_eval_verify(const std::vector<size_t>& prefix, int log_m, const std::vector<E>& z_row,
                                     const std::vector<E>& z_col, const std::vector<E>& z_trace, const SumcheckProof& proof,
                                     Challenger& ch, E* out) {
    const std::vector<E> col_eq = partial_lagrange(z_col);
    const E jagged_eval = proof.claimed_sum;
    ch.observe_ext(jagged_eval);
    if (prefix.empty()) return 10;
    const size_t dim = 2 * (size_t)(log_m + 1);
    if (int rc = partially_verify_sumcheck(proof, ch, dim, 2)) return 10 + rc;
    if (prefix.size() - 1 > col_eq.size()) return 10;
    const std::vector<E> first(proof.point.begin(), proof.point.begin() + dim / 2), second(proof.point.begin() + dim / 2, proof.point.end());
    E acc = E::zero();
    {
        std::vector<E> prev_merged;
        E prev_eval = E::zero();
        for (size_t c = 0; c + 1 < prefix.size(); c++) {
            std::vector<E> merged = point_from_usize(prefix[c], log_m + 1);
            const std::vector<E> nxt = point_from_usize(prefix[c + 1], log_m + 1);
            merged.insert(merged.end(), nxt.begin(), nxt.end());
            E fe;
            if (c > 0 && merged == prev_merged) fe = prev_eval;
            else { fe = full_lagrange_eval(merged, proof.point); prev_eval = fe; }
            prev_merged = merged;
            acc += col_eq[c] * fe;
        }
    }
    const BranchingProgram bp(z_row, z_trace);
    acc *= bp.eval(first, second);
    if (acc != proof.eval) return 15;
    *out = jagged_eval;
    return 0;
}

// ------------------------------------------------------------------ JaggedPcsProof
struct JaggedRoundData {                        // JaggedProverData of one commitment round
    std::vector<size_t> row_counts, column_counts;   // with the two padding tables appended
    size_t padding_column_count = 0;
    Digest original_commitment;                 // the stacked (inner) commitment
    std::vector<StackedBatch> batches;          // interleaved mles, each [2^lsh][w] row-major
    std::shared_ptr<BasefoldProverData> pcs;
};

// JaggedProver::commit_multilinears (prover.rs:L106-L160) keeping everything the evaluation proof needs.
// tables[k] = [rows_k][cols_k] row-major (rows_k may be 0: counted, not committed).
static inline Digest jagged_commit(const std::vector<TensorRef>& tables, int max_log_row_count, int log_stacking_height,
                                   size_t batch_size, const FriConfig& cfg, JaggedRoundData* out) {
    std::vector<TensorRef> dense;
    size_t area = 0;
    for (auto& t : tables) {
        out->row_counts.push_back(t.height);
        out->column_counts.push_back((size_t)t.width);
        if (t.height) { dense.push_back(t); area += t.height * (size_t)t.width; }
    }
    out->batches = interleave_fixed_rate(batch_size, dense, log_stacking_height);
    std::vector<MleRef> ms;
    for (auto& b : out->batches) ms.push_back(MleRef{b.data.data(), log_stacking_height, b.width});
    out->pcs = commit_mles(ms, cfg);
    out->original_commitment = out->pcs->tree.commit;
    const size_t H = (size_t)1 << log_stacking_height, M = (size_t)1 << max_log_row_count;
    const size_t added = std::max<size_t>((area + H - 1) / H, 1) * H - area;
    const size_t added_cols = std::max<size_t>((added + M - 1) / M, 1);
    out->row_counts.push_back(M);
    out->row_counts.push_back(added - (added_cols - 1) * M);
    out->column_counts.push_back(added_cols - 1);
    out->column_counts.push_back(1);
    out->padding_column_count = added_cols;
    std::vector<size_t> rows(out->row_counts.begin(), out->row_counts.end() - 2), cols(out->column_counts.begin(), out->column_counts.end() - 2);
    return jagged_commit_wrap(out->original_commitment, rows, cols, added, max_log_row_count);
}

struct JaggedProof {
    BasefoldProof basefold;
    std::vector<std::vector<E>> batch_evaluations;    // per round, all stacked columns
    SumcheckProof sumcheck, jagged_eval;
    std::vector<std::vector<std::pair<size_t, size_t>>> row_counts_and_column_counts;
    std::vector<Digest> merkle_tree_commitments;
    E expected_eval;
    size_t max_log_row_count = 0, log_m = 0;
};

static inline void write_basefold_proof(ByteWriter& w, const BasefoldProof& p) {
    std::vector<uint8_t> b = serialize_proof(p);
    w.b.insert(w.b.end(), b.begin(), b.end());
}

static inline std::vector<uint8_t> serialize_jagged_proof(const JaggedProof& p) {
    ByteWriter w;
    write_basefold_proof(w, p.basefold);
    w.u64(p.batch_evaluations.size());
    for (auto& r : p.batch_evaluations) { w.u64(r.size()); for (auto& e : r) w.e(e); w.u64(1); w.u64(r.size()); }
    write_sumcheck(w, p.sumcheck);
    write_sumcheck(w, p.jagged_eval);
    w.u64(p.row_counts_and_column_counts.size());
    for (auto& r : p.row_counts_and_column_counts) { w.u64(r.size()); for (auto& rc : r) { w.u64(rc.first); w.u64(rc.second); } }
    w.u64(p.merkle_tree_commitments.size());
    for (auto& c : p.merkle_tree_commitments) w.d(c);
    w.e(p.expected_eval);
    w.u64(p.max_log_row_count);
    w.u64(p.log_m);
    return w.b;
}

static inline JaggedProof deserialize_jagged_proof(const uint8_t* buf, size_t len) {
    ByteReader r{buf, len};
    JaggedProof p;
    p.basefold = read_basefold_proof(r);
    size_t n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    p.batch_evaluations.resize(n);
    for (auto& ev : p.batch_evaluations) {
        size_t k = r.u64();
        if (k > len) throw std::runtime_error("bad length");
        ev.resize(k);
        for (auto& e : ev) e = r.e();
        if (r.u64() != 1 || r.u64() != k) throw std::runtime_error("bad MleEval shape");
    }
    p.sumcheck = read_sumcheck(r);
    p.jagged_eval = read_sumcheck(r);
    n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    p.row_counts_and_column_counts.resize(n);
    for (auto& v : p.row_counts_and_column_counts) {
        size_t k = r.u64();
        if (k > len) throw std::runtime_error("bad length");
        v.resize(k);
        for (auto& rc : v) { rc.first = r.u64(); rc.second = r.u64(); }
    }
    n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    p.merkle_tree_commitments.resize(n);
    for (auto& c : p.merkle_tree_commitments) c = r.d();
    p.expected_eval = r.e();
    p.max_log_row_count = r.u64();
    p.log_m = r.u64();
    if (r.o != len) throw std::runtime_error("trailing bytes");
    return p;
}

// prover.rs:L162-L328. claims[r] = all column evaluations at z_row of round r's tables, in table order
// (without the padding columns).
static inline JaggedProof jagged_prove(const std::vector<E>& z_row, const std::vector<std::vector<E>>& claims,
                                       const std::vector<JaggedRoundData>& rounds, int max_log_row_count, int log_stacking_height,
                                       const FriConfig& cfg, Challenger& ch) {
    size_t total_cols = 0;
    for (auto& d : rounds) total_cols += std::accumulate(d.column_counts.begin(), d.column_counts.end(), (size_t)0);
    const int num_col_variables = log2_ceil_usize(total_cols);
    const std::vector<E> z_col = sample_point(ch, num_col_variables);

    std::vector<E> column_claims;
    for (size_t r = 0; r < rounds.size(); r++) {
        column_claims.insert(column_claims.end(), claims[r].begin(), claims[r].end());
        column_claims.insert(column_claims.end(), rounds[r].padding_column_count, E::zero());
    }
    std::vector<size_t> heights;
    for (auto& d : rounds)
        for (size_t t = 0; t < d.row_counts.size(); t++)
            for (size_t c = 0; c < d.column_counts[t]; c++) heights.push_back(d.row_counts[t]);
    const JaggedParams params = JaggedParams::from_column_heights(heights, max_log_row_count);
    const int log_m = params.log_m();

    // dense vector: every stacked column of every batch of every round, zero-padded to 2^log_m
    std::vector<F> q((size_t)1 << log_m, F::zero());
    {
        const size_t H = (size_t)1 << log_stacking_height;
        size_t off = 0;
        for (auto& d : rounds)
            for (auto& b : d.batches)
                for (int c = 0; c < b.width; c++) {
                    for (size_t i = 0; i < H; i++) q[off + i] = b.data[i * b.width + c];
                    off += H;
                }
    }
    const std::vector<E> jt = partial_jagged_table(params, z_row, z_col);

    column_claims.resize((size_t)1 << num_col_variables, E::zero());
    const E sumcheck_claim = eval_ext_mle_at_point(column_claims, z_col);

    JaggedProof proof;
    E q_eval, j_eval;
    proof.sumcheck = hadamard_sumcheck(q, jt, sumcheck_claim, ch, &q_eval, &j_eval);
    const std::vector<E>& final_point = proof.sumcheck.point;
    proof.jagged_eval = jagged_eval_prove(params, z_row, z_col, final_point, ch);
    proof.expected_eval = q_eval;

    // StackedPcsProver::prove_untrusted_evaluation
    ch.observe_ext(q_eval);
    const std::vector<E> stack_point = last_k(final_point, log_stacking_height);
    std::vector<std::vector<MleRef>> mle_rounds;
    std::vector<std::vector<std::vector<E>>> bf_claims;
    std::vector<std::shared_ptr<BasefoldProverData>> pdata;
    for (auto& d : rounds) {
        std::vector<MleRef> ms;
        std::vector<std::vector<E>> cs;
        std::vector<E> flat;
        for (auto& b : d.batches) {
            ms.push_back(MleRef{b.data.data(), log_stacking_height, b.width});
            cs.push_back(eval_mle_at_point(b.data.data(), (size_t)1 << log_stacking_height, b.width, stack_point));
            flat.insert(flat.end(), cs.back().begin(), cs.back().end());
        }
        mle_rounds.push_back(ms);
        bf_claims.push_back(cs);
        proof.batch_evaluations.push_back(flat);
        pdata.push_back(d.pcs);
    }
    for (auto& r : bf_claims) for (auto& m : r) for (auto& e : m) ch.observe_ext(e);
    proof.basefold = basefold_prove(stack_point, mle_rounds, bf_claims, pdata, cfg, ch);

    for (auto& d : rounds) {
        std::vector<std::pair<size_t, size_t>> rc;
        for (size_t t = 0; t < d.row_counts.size(); t++) rc.push_back({d.row_counts[t], d.column_counts[t]});
        proof.row_counts_and_column_counts.push_back(rc);
        proof.merkle_tree_commitments.push_back(d.original_commitment);
    }
    proof.max_log_row_count = max_log_row_count;
    proof.log_m = log_m;
    return proof;
}

// verifier.rs:L109-L383 + stacked/src/verifier.rs:L39-L99. commitments[r] = the jagged (outer) commitment
// of round r; claims[r] = that round's column evaluations at z_row. Returns 0 when the proof verifies.
static inline int jagged_verify(const std::vector<Digest>& commitments, const std::vector<E>& z_row,
                                const std::vector<std::vector<E>>& claims, const JaggedProof& proof, int max_log_row_count,
                                int log_stacking_height, const FriConfig& cfg, Challenger& ch) {
    const auto& rcs = proof.row_counts_and_column_counts;
    for (auto& r : rcs) if (r.empty()) return 1;
    // unzip_and_prefix_sums
    std::vector<std::vector<size_t>> row_counts, column_counts;
    std::vector<size_t> prefix{0};
    for (auto& r : rcs) {
        std::vector<size_t> rc, cc;
        for (auto& p : r) {
            rc.push_back(p.first); cc.push_back(p.second);
            for (size_t c = 0; c < p.second; c++) prefix.push_back(prefix.back() + p.first);
        }
        row_counts.push_back(rc); column_counts.push_back(cc);
    }
    const size_t purported_log_m = log2_ceil_usize(prefix.back());
    if (proof.max_log_row_count != (size_t)max_log_row_count || proof.log_m != purported_log_m) return 1;
    const int num_col_variables = log2_ceil_usize(prefix.size() - 1);
    const std::vector<E> z_col = sample_point(ch, num_col_variables);
    if (z_row.size() != (size_t)max_log_row_count) return 1;
    std::vector<E> column_claims;
    for (auto& c : claims) column_claims.insert(column_claims.end(), c.begin(), c.end());
    const size_t nr = commitments.size();
    if (claims.size() != nr || row_counts.size() != nr || proof.merkle_tree_commitments.size() != nr) return 1;
    for (size_t r = 0; r < nr; r++) if (row_counts[r].size() < 2) return 1;
    for (size_t r = 0; r < nr; r++) {
        size_t expected_len = 0;
        for (size_t t = 0; t + 2 < column_counts[r].size(); t++) expected_len += column_counts[r][t];
        if (claims[r].size() != expected_len) return 1;
    }
    for (size_t r = 0; r < nr; r++) {
        std::vector<F> in;
        in.push_back(F::from_canonical((uint32_t)column_counts[r].size()));
        for (size_t x : row_counts[r]) { if (x >= KB_P) return 2; in.push_back(F::from_canonical((uint32_t)x)); }
        for (size_t x : column_counts[r]) { if (x >= KB_P) return 2; in.push_back(F::from_canonical((uint32_t)x)); }
        if (compress(proof.merkle_tree_commitments[r], hash_slice(in.data(), in.size())) != commitments[r]) return 3;
    }
    const size_t M = (size_t)1 << max_log_row_count, H = (size_t)1 << log_stacking_height;
    std::vector<size_t> round_areas, added_cols;
    for (size_t r = 0; r < nr; r++) {
        size_t area = 0;
        for (size_t t = 0; t + 2 < row_counts[r].size(); t++) area += row_counts[r][t] * column_counts[r][t];
        if (area == 0 || area >= ((size_t)1 << 30)) return 4;
        const size_t added_vals = (area + H - 1) / H * H - area;
        const size_t exp_cols = std::max<size_t>((added_vals + M - 1) / M, 1);
        const auto& rc = row_counts[r];
        const auto& cc = column_counts[r];
        if (cc[cc.size() - 2] + 1 != exp_cols || cc.back() != 1 || rc[rc.size() - 2] != M ||
            rc.back() != added_vals - (exp_cols - 1) * M)
            return 1;
        for (size_t x : rc) if (x > M) return 1;
        round_areas.push_back(area + added_vals);
        added_cols.push_back(exp_cols);
    }
    if (proof.log_m >= 30) return 4;
    // insert the zero claims of the padding columns (from the last round backwards)
    {
        std::vector<size_t> insertion;
        size_t s = 0;
        for (size_t r = 0; r < nr; r++) { for (size_t t = 0; t + 2 < column_counts[r].size(); t++) s += column_counts[r][t]; insertion.push_back(s); }
        for (size_t r = nr; r-- > 0;) column_claims.insert(column_claims.begin() + insertion[r], added_cols[r], E::zero());
    }
    if (prefix.size() != column_claims.size() + 1) return 1;
    column_claims.resize((size_t)1 << num_col_variables, E::zero());
    const E sumcheck_claim = eval_ext_mle_at_point(column_claims, z_col);
    if (sumcheck_claim != proof.sumcheck.claimed_sum) return 5;
    const int log_trace = log2_ceil_usize(prefix.back());
    if (int rc = partially_verify_sumcheck(proof.sumcheck, ch, log_trace, 2)) return 20 + rc;
    for (size_t c = 0; c + 1 < prefix.size(); c++) if (prefix[c] > prefix[c + 1]) return 6;
    E jagged_eval;
    if (int rc = jagged_eval_verify(prefix, (int)proof.log_m, z_row, z_col, proof.sumcheck.point, proof.jagged_eval, ch, &jagged_eval))
        return 30 + rc;
    if (proof.expected_eval * jagged_eval != proof.sumcheck.eval) return 7;

    // stacked verify_untrusted_evaluation
    ch.observe_ext(proof.expected_eval);
    const std::vector<E>& point = proof.sumcheck.point;
    if (point.size() < (size_t)log_stacking_height) return 1;
    const std::vector<E> batch_point(point.begin(), point.end() - log_stacking_height), stack_point = last_k(point, log_stacking_height);
    if (proof.batch_evaluations.size() != nr) return 1;
    std::vector<E> all;
    for (size_t r = 0; r < nr; r++) {
        if (round_areas[r] % H || round_areas[r] / H != proof.batch_evaluations[r].size()) return 1;
        all.insert(all.end(), proof.batch_evaluations[r].begin(), proof.batch_evaluations[r].end());
    }
    if (all.size() > ((size_t)1 << batch_point.size())) return 1;
    all.resize((size_t)1 << batch_point.size(), E::zero());
    if (eval_ext_mle_at_point(all, batch_point) != proof.expected_eval) return 8;
    for (auto& r : proof.batch_evaluations) for (auto& e : r) ch.observe_ext(e);
    const BfError be = basefold_verify(proof.merkle_tree_commitments, stack_point, proof.batch_evaluations, proof.basefold, cfg, ch);
    return be == BfError::Ok ? 0 : 100 + (int)be;
}

}  // namespace orc
