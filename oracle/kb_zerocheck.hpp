// oracle/kb_zerocheck.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// CPU restatement of the zerocheck sumcheck over AIR constraints (SURVEY §8 rows a9–a12):
//   ZeroCheckPoly / VirtualGeq state        /root/reference/crates/hypercube/src/prover/zerocheck/mod.rs:L25-L48
//                                           /root/reference/slop/crates/multilinear/src/virtual_geq.rs:L12-L99
//   sum_as_poly_in_last_variable (y0,y2,y4) /root/reference/crates/hypercube/src/prover/zerocheck/sum_as_poly.rs:L53-L181
//   univariate assembly + interpolation     sum_as_poly.rs:L187-L287 ; /root/reference/slop/crates/algebra/src/univariate.rs:L85-L108
//   increment_y_values / folder             sum_as_poly.rs:L355-L440 ; /root/reference/crates/hypercube/src/folder.rs:L276-L323
//   zerocheck_fix_last_variable             /root/reference/crates/hypercube/src/prover/zerocheck/fix_last_variable.rs:L8-L62
//   mle_fix_last_variable                   /root/reference/slop/crates/multilinear/src/restrict.rs:L11-L58
//   driver ShardProver::zerocheck           /root/reference/crates/hypercube/src/prover/shard.rs:L474-L646
//   reduce_sumcheck_to_evaluation           /root/reference/slop/crates/sumcheck/src/prover.rs:L13-L96
//   verifier equation (used as the check)   /root/reference/crates/hypercube/src/verifier/shard.rs:L288-L435
//   partially_verify_sumcheck_proof         /root/reference/slop/crates/sumcheck/src/verifier.rs:L22-L98
//
// Constraints are DATA here (the reference's chips are Rust generic code over `AirBuilder`): an SSA
// program over {load main/preprocessed column, constant, public value, add, sub, mul, neg,
// assert_zero}, the same operation set as the reference's own GPU bytecode
// (/root/reference/sp1-gpu/crates/sys/include/zerocheck/sequential.cuh:L1-L149). Single-row AIRs only,
// as `ConstraintSumcheckFolder` (no next-row access).
// Pinned by: prover -> verifier-equation round trips (the reference's own test style) and the sumcheck
// round-consistency of the reference's real zerocheck proof (tests/golden). The RISC-V chips'
// constraints themselves are not available without a Rust toolchain ("parity unpinned" for them).
#pragma once
#include <array>
#include <map>

#include "kb_pcs.hpp"

namespace orc {

enum ZcOp : uint32_t { ZC_LOAD_MAIN = 0, ZC_LOAD_PREP = 1, ZC_CONST = 2, ZC_PUBLIC = 3, ZC_ADD = 4, ZC_SUB = 5, ZC_MUL = 6,
                       ZC_NEG = 7, ZC_ASSERT_ZERO = 8,
                       ZC_HINT = 16 };   // pseudo-instruction (a fused-kernel hint for provers): defines no value, asserts nothing

struct ZcInstr {
    uint32_t op, a, b;   // SSA: the value of instruction k is register k (ASSERT_ZERO defines none)
};

struct ZcAir {
    std::vector<ZcInstr> prog;
    int main_width = 0, prep_width = 0, num_constraints = 0;
};

template <class K> struct KOps;
template <> struct KOps<F> {
    static F from_f(F x) { return x; }
    static E scale(const E& a, F k) { return a * k; }
};
template <> struct KOps<E> {
    static E from_f(F x) { return E::from_base(x); }
    static E scale(const E& a, const E& k) { return a * k; }
};

// air.eval(&mut ConstraintSumcheckFolder): accumulator += powers_of_alpha[constraint_index++] * x
template <class K>
static inline E eval_constraints(const ZcAir& air, const K* prep, const K* main, const F* publics, const E* alpha_pows) {
    std::vector<K> reg(air.prog.size());
    E acc = E::zero();
    int ci = 0;
    for (size_t k = 0; k < air.prog.size(); k++) {
        const ZcInstr& in = air.prog[k];
        switch (in.op) {
            case ZC_LOAD_MAIN: reg[k] = main[in.a]; break;
            case ZC_LOAD_PREP: reg[k] = prep[in.a]; break;
            case ZC_CONST: reg[k] = KOps<K>::from_f(F::from_canonical(in.a)); break;
            case ZC_PUBLIC: reg[k] = KOps<K>::from_f(publics[in.a]); break;
            case ZC_ADD: reg[k] = reg[in.a] + reg[in.b]; break;
            case ZC_SUB: reg[k] = reg[in.a] - reg[in.b]; break;
            case ZC_MUL: reg[k] = reg[in.a] * reg[in.b]; break;
            case ZC_NEG: reg[k] = -reg[in.a]; break;
            case ZC_ASSERT_ZERO: acc += KOps<K>::scale(alpha_pows[ci++], reg[in.a]); break;
            case ZC_HINT: break;
            default: throw std::runtime_error("bad zerocheck opcode");
        }
    }
    return acc;
}

// Verifier-side folding (`VerifierConstraintFolder`): Horner in alpha, acc = acc * alpha + x.
static inline E eval_constraints_horner(const ZcAir& air, const E* prep, const E* main, const F* publics, const E& alpha) {
    std::vector<E> pows(air.num_constraints);
    E cur = E::one();
    for (int i = 0; i < air.num_constraints; i++) { pows[air.num_constraints - 1 - i] = cur; cur *= alpha; }
    return eval_constraints<E>(air, prep, main, publics, pows.data());
}

struct VirtualGeq {
    uint32_t threshold;
    E geq_coefficient, eq_coefficient;
    uint32_t num_vars;
    VirtualGeq fix_last_variable(const E& alpha) const {
        VirtualGeq r;
        r.threshold = threshold >> 1;
        r.geq_coefficient = geq_coefficient;
        r.eq_coefficient = (threshold & 1) == 0 ? (E::one() - alpha) * eq_coefficient
                                                : alpha * (eq_coefficient + geq_coefficient) - geq_coefficient;
        r.num_vars = num_vars ? num_vars - 1 : 0;
        return r;
    }
    E eval_at_usize(size_t index) const {
        if (index < threshold) return E::zero();
        if (index == threshold) return eq_coefficient + geq_coefficient;
        return geq_coefficient;
    }
};

using UniPoly = std::vector<E>;

static inline E uni_eval(const UniPoly& p, const E& x) {
    E acc = E::zero();
    for (size_t i = p.size(); i-- > 0;) acc = acc * x + p[i];
    return acc;
}
static inline E uni_eval_one_plus_eval_zero(const UniPoly& p) {
    if (p.empty()) return E::zero();
    E s = p[0];
    for (auto& c : p) s += c;
    return s;
}
static inline UniPoly uni_add(const UniPoly& a, const UniPoly& b) {
    UniPoly r(std::max(a.size(), b.size()), E::zero());
    for (size_t i = 0; i < r.size(); i++) r[i] = (i < a.size() ? a[i] : E::zero()) + (i < b.size() ? b[i] : E::zero());
    return r;
}
static inline UniPoly uni_scale(UniPoly a, const E& k) { for (auto& c : a) c *= k; return a; }
static inline UniPoly uni_mul_by_x(const UniPoly& a) { UniPoly r{E::zero()}; r.insert(r.end(), a.begin(), a.end()); return r; }

static inline UniPoly interpolate_univariate(const std::vector<E>& xs, const std::vector<E>& ys) {
    UniPoly result{E::zero()};
    for (size_t i = 0; i < xs.size(); i++) {
        E den = E::one();
        UniPoly num{ys[i]};
        for (size_t j = 0; j < xs.size(); j++) {
            if (j == i) continue;
            den *= xs[i] - xs[j];
            num = uni_add(uni_mul_by_x(num), uni_scale(num, -xs[j]));
        }
        result = uni_add(result, uni_scale(num, einv(den)));
    }
    return result;
}
static inline UniPoly rlc_univariate(const std::vector<UniPoly>& polys, const E& lambda) {
    UniPoly result{E::zero()};
    for (auto& p : polys) result = uni_add(uni_scale(result, lambda), p);
    return result;
}

// One chip's zerocheck polynomial, generic over the trace element type of the current round.
template <class K>
struct ZcPoly {
    const ZcAir* air;
    const F* publics;
    std::vector<E> alpha_pows, gkr_pows;
    std::vector<E> zeta;
    std::vector<K> prep, main;      // row-major [real_rows][width]
    size_t real_rows = 0;           // num_real_entries
    uint32_t num_vars = 0;
    E eq_adjustment, geq_value, padded_row_adjustment;
    VirtualGeq vgeq;
};

template <class K, bool FIRST>
static inline std::array<E, 3> zc_sum_y(const ZcPoly<K>& p, const std::vector<E>& eq) {
    const int w = p.air->main_width, wp = p.air->prep_width;
    const size_t terms = (p.real_rows + 1) / 2;
    E y0 = E::zero(), y2 = E::zero(), y4 = E::zero();
#pragma omp parallel
    {
        E l0 = E::zero(), l2 = E::zero(), l4 = E::zero();
        std::vector<K> m0(w), m2(w), m4(w), q0(wp), q2(wp), q4(wp);
        const K zero = KOps<K>::from_f(F::zero());
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < terms; i++) {
            auto lerp = [&](const std::vector<K>& src, int width, std::vector<K>& v0, std::vector<K>& v2, std::vector<K>& v4) {
                for (int c = 0; c < width; c++) {
                    K r0 = src[(2 * i) * width + c];
                    K r1 = (2 * i + 1 < p.real_rows) ? src[(2 * i + 1) * width + c] : zero;
                    K slope = r1 - r0, s2 = slope + slope, s4 = s2 + s2;
                    v0[c] = r0; v2[c] = s2 + r0; v4[c] = s4 + r0;
                }
            };
            lerp(p.main, w, m0, m2, m4);
            if (wp) lerp(p.prep, wp, q0, q2, q4);
            auto gkr = [&](const std::vector<K>& mv, const std::vector<K>& qv) {
                E g = E::zero();
                size_t k = 0;
                for (int c = 0; c < w && k < p.gkr_pows.size(); c++, k++) g += KOps<K>::scale(p.gkr_pows[k], mv[c]);
                for (int c = 0; c < wp && k < p.gkr_pows.size(); c++, k++) g += KOps<K>::scale(p.gkr_pows[k], qv[c]);
                return g;
            };
            E g0 = gkr(m0, q0), g2 = gkr(m2, q2), g4 = g2 + g2 - g0;
            E a0 = g0;
            if (!FIRST) a0 += eval_constraints<K>(*p.air, q0.data(), m0.data(), p.publics, p.alpha_pows.data());
            E a2 = eval_constraints<K>(*p.air, q2.data(), m2.data(), p.publics, p.alpha_pows.data()) + g2;
            E a4 = eval_constraints<K>(*p.air, q4.data(), m4.data(), p.publics, p.alpha_pows.data()) + g4;
            l0 += a0 * eq[i]; l2 += a2 * eq[i]; l4 += a4 * eq[i];
        }
#pragma omp critical
        { y0 += l0; y2 += l2; y4 += l4; }
    }
    return {y0, y2, y4};
}

template <class K, bool FIRST>
static inline UniPoly zc_sum_as_poly(const ZcPoly<K>& p, const E& claim) {
    if (p.real_rows == 0) return UniPoly(5, E::zero());
    std::vector<E> rest(p.zeta.begin(), p.zeta.end() - 1);
    const E last = p.zeta.back();
    std::vector<E> eq = partial_lagrange(rest);
    auto y = zc_sum_y<K, FIRST>(p, eq);
    const size_t threshold_half = (p.real_rows + 1) / 2 - 1;
    E msb = threshold_half < ((size_t)1 << (p.num_vars - 1)) ? p.eq_adjustment * eq[threshold_half] : E::zero();
    const E four = E::from_base(F::from_canonical(4)), two = E::from_base(F::two());
    E v0 = p.vgeq.fix_last_variable(E::zero()).eval_at_usize(threshold_half);
    E v2 = p.vgeq.fix_last_variable(two).eval_at_usize(threshold_half);
    E v4 = p.vgeq.fix_last_variable(four).eval_at_usize(threshold_half);
    std::vector<E> xs, ys;
    E f0 = E::one() - last;
    E y0 = y[0] * (f0 * p.eq_adjustment) - p.padded_row_adjustment * v0 * msb * f0;
    xs.push_back(E::zero()); ys.push_back(y0);
    xs.push_back(E::one()); ys.push_back(claim - y0);
    E f2 = last * F::from_canonical(3) - E::one();
    E y2 = y[1] * (f2 * p.eq_adjustment) - p.padded_row_adjustment * v2 * msb * f2;
    xs.push_back(two); ys.push_back(y2);
    E f4 = last * F::from_canonical(7) - E::from_base(F::from_canonical(3));
    E y4 = y[2] * (f4 * p.eq_adjustment) - p.padded_row_adjustment * v4 * msb * f4;
    xs.push_back(four); ys.push_back(y4);
    E b = (E::one() - last) * einv(E::one() - (last + last));
    xs.push_back(b); ys.push_back(E::zero());
    return interpolate_univariate(xs, ys);
}

template <class K>
static inline std::vector<E> zc_fix_rows(const std::vector<K>& src, size_t real_rows, int width, const E& alpha) {
    const size_t out_rows = (real_rows + 1) / 2;
    std::vector<E> out(out_rows * width);
    const K zero = KOps<K>::from_f(F::zero());
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < out_rows; i++)
        for (int c = 0; c < width; c++) {
            K x = src[(2 * i) * width + c];
            K y = (2 * i + 1 < real_rows) ? src[(2 * i + 1) * width + c] : zero;
            out[i * width + c] = KOps<K>::scale(alpha, y - x) + x;
        }
    return out;
}

template <class K>
static inline ZcPoly<E> zc_fix_last_variable(const ZcPoly<K>& p, const E& alpha) {
    ZcPoly<E> r;
    r.air = p.air; r.publics = p.publics; r.alpha_pows = p.alpha_pows; r.gkr_pows = p.gkr_pows;
    r.prep = zc_fix_rows<K>(p.prep, p.real_rows, p.air->prep_width, alpha);
    r.main = zc_fix_rows<K>(p.main, p.real_rows, p.air->main_width, alpha);
    r.real_rows = (p.real_rows + 1) / 2;
    r.num_vars = p.num_vars - 1;
    r.padded_row_adjustment = p.padded_row_adjustment;
    r.vgeq = p.vgeq.fix_last_variable(alpha);
    if (p.real_rows == 0) {
        r.zeta = p.zeta;               // pure padding: nothing else is propagated (fix_last_variable.rs:L21-L34)
        r.eq_adjustment = p.eq_adjustment;
        r.geq_value = p.geq_value;
        return r;
    }
    const E last = p.zeta.back();
    r.zeta.assign(p.zeta.begin(), p.zeta.end() - 1);
    r.eq_adjustment = p.eq_adjustment * (alpha * last + (E::one() - alpha) * (E::one() - last));
    r.geq_value = p.real_rows > 1 ? E::zero() : (E::one() - p.geq_value) * alpha + p.geq_value;
    return r;
}

struct ZcChipInput {
    const ZcAir* air;
    const F* main;        // [real_rows][main_width] row-major
    const F* prep;        // [real_rows][prep_width] or null
    size_t real_rows;
    std::vector<E> main_opening, prep_opening;   // trace column evaluations at zeta (from LogUp-GKR)
};

struct ZcProof {
    std::vector<UniPoly> univariate_polys;
    E claimed_sum;
    std::vector<E> point;
    E eval;
    std::vector<std::vector<E>> chip_evals;  // per chip: preprocessed columns then main columns
};

static inline ZcProof zerocheck_prove(const std::vector<ZcChipInput>& chips, int max_log_row_count, const std::vector<E>& zeta,
                                      const E& batching_challenge, const E& gkr_batch, const std::vector<F>& publics,
                                      Challenger& ch) {
    int max_constraints = 0;
    for (auto& c : chips) max_constraints = std::max(max_constraints, c.air->num_constraints);
    std::vector<E> pows(max_constraints);
    { E cur = E::one(); for (auto& x : pows) { x = cur; cur *= batching_challenge; } }
    std::vector<ZcPoly<F>> polys;
    std::vector<E> claims;
    for (auto& c : chips) {
        ZcPoly<F> p;
        p.air = c.air; p.publics = publics.data();
        p.alpha_pows.assign(pows.begin(), pows.begin() + c.air->num_constraints);
        std::reverse(p.alpha_pows.begin(), p.alpha_pows.end());
        std::vector<F> zm(c.air->main_width, F::zero()), zp(c.air->prep_width, F::zero());
        p.padded_row_adjustment = eval_constraints<F>(*c.air, zp.data(), zm.data(), publics.data(), p.alpha_pows.data());
        { E cur = gkr_batch; for (int i = 0; i < c.air->main_width + c.air->prep_width; i++) { p.gkr_pows.push_back(cur); cur *= gkr_batch; } }
        E claim = E::zero();
        { size_t k = 0;
          for (auto& o : c.main_opening) claim += o * p.gkr_pows[k++];
          for (auto& o : c.prep_opening) claim += o * p.gkr_pows[k++]; }
        p.zeta = zeta;
        p.main.assign(c.main, c.main + c.real_rows * c.air->main_width);
        if (c.air->prep_width) p.prep.assign(c.prep, c.prep + c.real_rows * c.air->prep_width);
        p.real_rows = c.real_rows;
        p.num_vars = max_log_row_count;
        p.eq_adjustment = E::one();
        p.geq_value = c.real_rows > 0 ? E::zero() : E::one();
        p.vgeq = VirtualGeq{(uint32_t)c.real_rows, E::one(), E::zero(), (uint32_t)max_log_row_count};
        polys.push_back(std::move(p));
        claims.push_back(claim);
    }
    const E lambda = ch.sample_ext();
    ZcProof proof;
    std::vector<E> point;
    std::vector<UniPoly> uni(polys.size());
    for (size_t i = 0; i < polys.size(); i++) uni[i] = zc_sum_as_poly<F, true>(polys[i], claims[i]);
    UniPoly rlc = rlc_univariate(uni, lambda);
    for (auto& c : rlc) ch.observe_ext(c);
    proof.univariate_polys.push_back(rlc);
    E alpha = ch.sample_ext();
    point.insert(point.begin(), alpha);
    std::vector<ZcPoly<E>> cur;
    for (auto& p : polys) cur.push_back(zc_fix_last_variable<F>(p, alpha));
    for (int r = 1; r < max_log_row_count; r++) {
        std::vector<E> round_claims;
        for (auto& u : uni) round_claims.push_back(uni_eval(u, point.front()));
        for (size_t i = 0; i < cur.size(); i++) uni[i] = zc_sum_as_poly<E, false>(cur[i], round_claims[i]);
        rlc = rlc_univariate(uni, lambda);
        for (auto& c : rlc) ch.observe_ext(c);
        proof.univariate_polys.push_back(rlc);
        alpha = ch.sample_ext();
        point.insert(point.begin(), alpha);
        std::vector<ZcPoly<E>> nxt;
        for (auto& p : cur) nxt.push_back(zc_fix_last_variable<E>(p, alpha));
        cur.swap(nxt);
    }
    proof.claimed_sum = E::zero();
    for (auto& c : claims) proof.claimed_sum = proof.claimed_sum * lambda + c;
    proof.point = point;
    proof.eval = E::zero();
    for (auto& u : uni) proof.eval = proof.eval * lambda + uni_eval(u, point.front());
    // component poly evals: preprocessed then main (mod.rs:L95-L117); pure-padding chips give zeros
    for (auto& p : cur) {
        std::vector<E> ev;
        for (int c = 0; c < p.air->prep_width; c++) ev.push_back(p.real_rows ? p.prep[c] : E::zero());
        for (int c = 0; c < p.air->main_width; c++) ev.push_back(p.real_rows ? p.main[c] : E::zero());
        proof.chip_evals.push_back(ev);
    }
    // observe the openings (shard.rs:L609-L640)
    ch.observe(F::from_canonical((uint32_t)chips.size()));
    for (size_t i = 0; i < chips.size(); i++) {
        const int wp = chips[i].air->prep_width;
        ch.observe(F::from_canonical((uint32_t)wp));
        for (int c = 0; c < wp; c++) ch.observe_ext(proof.chip_evals[i][c]);
        ch.observe(F::from_canonical((uint32_t)chips[i].air->main_width));
        for (int c = 0; c < chips[i].air->main_width; c++) ch.observe_ext(proof.chip_evals[i][wp + c]);
    }
    return proof;
}

static inline E full_lagrange_eval(const std::vector<E>& a, const std::vector<E>& b) {
    E acc = E::one();
    for (size_t i = 0; i < a.size(); i++) acc *= a[i] * b[i] + (E::one() - a[i]) * (E::one() - b[i]);
    return acc;
}
// full_geq(threshold bits (big-endian), point) — mle.rs:L398-L407
static inline E full_geq(const std::vector<F>& threshold, const std::vector<E>& point) {
    E acc = E::one();
    for (size_t k = threshold.size(); k-- > 0;) {
        const F x = threshold[k];
        const E& y = point[k];
        acc = ((E::one() - y) * (F::one() - x) + y * x) * acc + y * (F::one() - x);
    }
    return acc;
}

// Returns 0 when the proof satisfies the reference verifier's zerocheck checks, else a code 1..5.
static inline int zerocheck_verify(const std::vector<const ZcAir*>& airs, const std::vector<size_t>& heights,
                                   const std::vector<std::vector<E>>& main_openings_gkr,
                                   const std::vector<std::vector<E>>& prep_openings_gkr, int max_log_row_count,
                                   const std::vector<E>& zeta, const E& alpha, const E& gkr_batch, const std::vector<F>& publics,
                                   const ZcProof& proof, Challenger& ch) {
    const E lambda = ch.sample_ext();
    if ((int)zeta.size() != max_log_row_count || (int)proof.point.size() != max_log_row_count) return 1;
    const E eq_val = full_lagrange_eval(zeta, proof.point);
    E rlc = E::zero();
    for (size_t i = 0; i < airs.size(); i++) {
        const ZcAir& air = *airs[i];
        std::vector<E> ext_point = proof.point;
        ext_point.insert(ext_point.begin(), E::zero());
        std::vector<F> degree(max_log_row_count + 1);
        for (int k = 0; k <= max_log_row_count; k++) degree[k] = F::from_canonical((uint32_t)((heights[i] >> (max_log_row_count - k)) & 1));
        const E geq_val = full_geq(degree, ext_point);
        std::vector<E> zm(air.main_width, E::zero()), zp(air.prep_width, E::zero());
        const E pad_adj = eval_constraints_horner(air, zp.data(), zm.data(), publics.data(), alpha);
        const E* prep = proof.chip_evals[i].data();
        const E* main = prep + air.prep_width;
        const E constraint_eval = eval_constraints_horner(air, prep, main, publics.data(), alpha) - pad_adj * geq_val;
        E batch = E::zero(), pw = gkr_batch;
        for (int c = 0; c < air.main_width; c++) { batch += main[c] * pw; pw *= gkr_batch; }
        for (int c = 0; c < air.prep_width; c++) { batch += prep[c] * pw; pw *= gkr_batch; }
        rlc = rlc * lambda + eq_val * (constraint_eval + batch);
    }
    if (proof.eval != rlc) return 2;
    E mod = E::zero();
    for (size_t i = 0; i < airs.size(); i++) {
        E s = E::zero(), pw = gkr_batch;
        for (auto& o : main_openings_gkr[i]) { s += o * pw; pw *= gkr_batch; }
        for (auto& o : prep_openings_gkr[i]) { s += o * pw; pw *= gkr_batch; }
        mod = lambda * mod + s;
    }
    if (proof.claimed_sum != mod) return 3;
    // partially_verify_sumcheck_proof, degree 4
    if ((int)proof.univariate_polys.size() != max_log_row_count) return 1;
    const UniPoly* prev = &proof.univariate_polys[0];
    if (uni_eval_one_plus_eval_zero(*prev) != proof.claimed_sum) return 3;
    if (prev->size() != 5) return 1;
    for (auto& c : *prev) ch.observe_ext(c);
    std::vector<E> alpha_point;
    for (size_t r = 1; r < proof.univariate_polys.size(); r++) {
        const UniPoly& poly = proof.univariate_polys[r];
        if (poly.size() != 5) return 1;
        E a = ch.sample_ext();
        alpha_point.insert(alpha_point.begin(), a);
        if (uni_eval(*prev, a) != uni_eval_one_plus_eval_zero(poly)) return 4;
        for (auto& c : poly) ch.observe_ext(c);
        prev = &poly;
    }
    E a = ch.sample_ext();
    alpha_point.insert(alpha_point.begin(), a);
    if (alpha_point != proof.point) return 1;
    if (uni_eval(*prev, a) != proof.eval) return 5;
    return 0;
}

// bincode of PartialSumcheckProof<EF> followed by the per-chip opened values (u64 count, then per chip
// Vec<EF> preprocessed, Vec<EF> main) — /root/reference/slop/crates/sumcheck/src/proof.rs:L10-L14
static inline std::vector<uint8_t> serialize_zc_proof(const ZcProof& p) {
    ByteWriter w;
    w.u64(p.univariate_polys.size());
    for (auto& u : p.univariate_polys) { w.u64(u.size()); for (auto& c : u) w.e(c); }
    w.e(p.claimed_sum);
    w.u64(p.point.size());
    for (auto& x : p.point) w.e(x);
    w.e(p.eval);
    w.u64(p.chip_evals.size());
    for (auto& ev : p.chip_evals) { w.u64(ev.size()); for (auto& x : ev) w.e(x); }
    return w.b;
}

}  // namespace orc
