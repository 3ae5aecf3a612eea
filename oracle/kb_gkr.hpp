// oracle/kb_gkr.hpp — TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
//
// LogUp-GKR: the lookup argument between `commit_traces` and zerocheck. Restates
//   GkrProverImpl::prove_logup_gkr / prove_gkr_circuit   /root/reference/crates/hypercube/src/logup_gkr/prover.rs:L32-L215
//   generate_interaction_vals / generate_first_layer /
//   layer_transition / extract_outputs                    /root/reference/crates/hypercube/src/logup_gkr/execution.rs:L13-L382
//   prove_gkr_round                                        /root/reference/crates/hypercube/src/logup_gkr/cpu.rs:L146-L226
//   LogupRoundPolynomial (sum_as_poly / fix_t_variables)   /root/reference/crates/hypercube/src/logup_gkr/logup_poly.rs:L70-L553
//   Interaction::eval                                      /root/reference/crates/hypercube/src/lookup/interaction.rs:L171-L205
//   LogUpGkrVerifier::verify_logup_gkr                     /root/reference/crates/hypercube/src/logup_gkr/verifier.rs:L102-L354
//   proof structs (bincode order)                          /root/reference/crates/hypercube/src/logup_gkr/proof.rs:L8-L62
//
// Formulation. The reference keeps each chip's layer as PaddedMle's over its real rows and corrects for the
// padding rows (numerator 0, denominator 1) in closed form. This oracle instead materialises every layer
// DENSELY over all 2^v rows x 2^niv interactions with the padding values written out, and runs a plain
// dense sumcheck on five tables (n0, d0, n1, d1, eq). The round polynomial is the same mathematical object
// (degree 3, sent as the coefficients of the interpolation through p(0), p(1), p(1/2) and the root of the
// eq factor), field arithmetic is exact, so the messages are identical — and the GPU implementation, which
// works on real rows only with the closed-form correction like the reference, is checked against an
// independent algorithm. Small sizes only.
//
// Pinned by reference data as far as a proof allows: the verifier below (transcript order, round
// equations, output checks) accepts the reference's REAL LogupGkrProof from the replayed transcript
// (tests/test_oracle_golden.py); the final interaction check needs the machine's chips and is exercised on
// hand-written interactions only.
#pragma once
#include <map>
#include <string>
#include <tuple>

#include "kb_jagged.hpp"

namespace orc {

constexpr int GKR_GRINDING_BITS = 12;

struct VCol {                                   // VirtualPairCol: sum weight * column + constant
    std::vector<std::tuple<int, int, F>> terms; // (1 = main / 0 = preprocessed, column, weight)
    F constant = F::zero();
    template <class K>
    K apply(const K* prep, const K* main) const {
        K out = zero_of<K>();
        for (auto& t : terms) out = out + (std::get<0>(t) ? main : prep)[std::get<1>(t)] * std::get<2>(t);
        return out + constant;
    }
    template <class K> static K zero_of();
};
template <> inline F VCol::zero_of<F>() { return F::zero(); }
template <> inline E VCol::zero_of<E>() { return E::zero(); }

struct GkrInteraction {
    bool is_send = true;
    uint32_t kind = 0;                          // InteractionKind as usize (argument_index)
    VCol multiplicity;
    std::vector<VCol> values;
};

struct GkrChip {                                // chips must be given in name order (BTreeSet<Chip>)
    std::string name;
    std::vector<GkrInteraction> interactions;   // sends first, then receives (cpu.rs:L86-L92)
    int main_width = 0, prep_width = 0;
    const F* main = nullptr;                    // [real_rows][main_width] row-major
    const F* prep = nullptr;                    // [real_rows][prep_width] or null
    size_t real_rows = 0;
};

struct GkrRoundProof { E numerator_0, numerator_1, denominator_0, denominator_1; SumcheckProof sumcheck; };

struct GkrProof {
    std::vector<E> numerator, denominator;      // circuit output, 2^(niv + 1) each
    std::vector<GkrRoundProof> rounds;
    std::vector<E> point;                       // LogUpEvaluations.point (the trace point)
    std::vector<std::string> chip_names;
    std::vector<std::vector<E>> main_evals;
    std::vector<std::vector<E>> prep_evals;     // empty vector + has_prep false when the chip has none
    std::vector<bool> has_prep;
    F witness;
};

static inline void interaction_vals(const GkrInteraction& in, const F* prep, const F* main, const E& alpha, const std::vector<E>& betas,
                                    F* mult, E* denom) {
    E d = alpha + betas[0] * F::from_canonical(in.kind);
    for (size_t j = 0; j < in.values.size(); j++) d += betas[1 + j] * in.values[j].apply<F>(prep, main);
    F m = in.multiplicity.apply<F>(prep, main);
    *mult = in.is_send ? m : -m;
    *denom = d;
}

// Interaction::eval on opened (extension) values
static inline void interaction_eval_ext(const GkrInteraction& in, const E* prep, const E* main, const E& alpha, const std::vector<E>& betas,
                                        E* mult, E* fingerprint) {
    *mult = in.multiplicity.apply<E>(prep, main);
    E f = alpha + betas[0] * F::from_canonical(in.kind);
    for (size_t j = 0; j < in.values.size(); j++) f += in.values[j].apply<E>(prep, main) * betas[1 + j];
    *fingerprint = f;
}

struct FracTable {                               // [rows][width] of (N, D)
    size_t rows = 0, width = 0;
    std::vector<E> n, d;
};

static inline int gkr_beta_seed_dim(const std::vector<GkrChip>& chips) {
    size_t arity = 0;
    for (auto& c : chips) for (auto& i : c.interactions) arity = std::max(arity, i.values.size() + 1);
    return log2_ceil(arity);
}

// `MachineRecord::eval_public_values` of a machine as data (sp1_amd/machines/public_values.py builds the RISC-V record's,
// /root/reference/crates/core/executor/src/record.rs:L879-L906): constraints over the public words alone and the record's own
// sends / receives, evaluated on the "row" of public values. A machine without one (the recursion machine: the trait's
// default body is empty) has no constraints, no interactions and `interactions_in_public_values()` = [].
struct PvProgram {
    ZcAir air;                                   // main_width = prep_width = 0; loads are ZC_PUBLIC / ZC_CONST
    std::vector<GkrInteraction> interactions;    // VCol "main" columns index the public values
    int num_pv_elts = 0;                         // machine.num_pv_elts(): words beyond it must be zero (verifier/shard.rs:L455-L464)
    int proof_max_num_pvs = 0;                   // PROOF_MAX_NUM_PVS, the length a ShardProof carries (0 = unchecked)
    size_t max_kind_arity = 1;                   // max over interactions_in_public_values() of num_values + 1 (verifier.rs:L120-L124)
};

// LogUpGkrVerifier::verify_public_values (verifier.rs:L74-L96): the constraints folded with `challenge` must vanish; returns
// the local interaction digest sum_sends m / (alpha + betas . (kind, values)) - sum_receives (folder.rs:L598-L620).
static inline bool verify_public_values(const PvProgram& pvp, const E& challenge, const E& alpha, const std::vector<E>& betas,
                                        const std::vector<F>& publics, E* digest) {
    if (eval_constraints_horner(pvp.air, nullptr, nullptr, publics.data(), challenge) != E::zero()) return false;
    E acc = E::zero();
    for (auto& in : pvp.interactions) {
        if (1 + in.values.size() > betas.size()) throw std::runtime_error("public-values interaction wider than the beta table");
        F m; E d;
        interaction_vals(in, nullptr, publics.data(), alpha, betas, &m, &d);     // m carries the sign: + send, - receive
        acc += E::from_base(m) * einv(d);
    }
    *digest = acc;
    return true;
}

// the rounds of a dense degree-3 sumcheck of sum_x eq[x] (lambda (n0 d1 + n1 d0) + d0 d1)[x] over 2^|pts| entries, binding the
// LAST variable first; pts[r] = coordinate of the variable round r binds (the root of the eq factor needs it). eq carries
// every factor already bound. Tables and claim are updated in place; messages / challenges are appended.
static inline void gkr_dense_rounds(std::vector<E>& n0, std::vector<E>& d0, std::vector<E>& n1, std::vector<E>& d1, std::vector<E>& eq,
                                    const std::vector<E>& pts, const E& lambda, E& claim, Challenger& ch, std::vector<UniPoly>& polys,
                                    std::vector<E>& alphas) {
    const E one = E::one(), two = E::from_base(F::two());
    const E inv2 = einv(two), inv8 = einv(E::from_base(F::from_canonical(8)));
    for (size_t r = 0; r < pts.size(); r++) {
        const size_t half = eq.size() / 2;
        E p0 = E::zero(), ph = E::zero();
#pragma omp parallel
        {
            E l0 = E::zero(), lh = E::zero();
#pragma omp for schedule(static) nowait
            for (size_t k = 0; k < half; k++) {
                const size_t a = 2 * k, b = 2 * k + 1;
                l0 += eq[a] * (lambda * (n0[a] * d1[a] + n1[a] * d0[a]) + d0[a] * d1[a]);
                const E sn0 = n0[a] + n0[b], sn1 = n1[a] + n1[b], sd0 = d0[a] + d0[b], sd1 = d1[a] + d1[b];
                lh += (eq[a] + eq[b]) * (lambda * (sn0 * sd1 + sn1 * sd0) + sd0 * sd1);
            }
#pragma omp critical
            { p0 += l0; ph += lh; }
        }
        ph = ph * inv8;
        const E pt = pts[r];
        const E b_const = (one - pt) * einv(one - (pt + pt));
        UniPoly uni = interpolate_univariate({E::zero(), one, inv2, b_const}, {p0, claim - p0, ph, E::zero()});
        for (auto& c : uni) ch.observe_ext(c);
        polys.push_back(uni);
        const E alpha = ch.sample_ext();
        alphas.push_back(alpha);
        claim = uni_eval(uni, alpha);
        n0 = fix_last_variable(n0, alpha); d0 = fix_last_variable(d0, alpha);
        n1 = fix_last_variable(n1, alpha); d1 = fix_last_variable(d1, alpha);
        eq = fix_last_variable(eq, alpha);
    }
}

// one GKR round on dense tables; tables are indexed idx = interaction * 2^v + row (row LSB = last variable)
static inline GkrRoundProof gkr_round_dense(std::vector<E> n0, std::vector<E> d0, std::vector<E> n1, std::vector<E> d1,
                                            const std::vector<E>& eval_point, const E& num_eval, const E& den_eval, Challenger& ch) {
    const E lambda = ch.sample_ext();
    std::vector<E> eq = partial_lagrange(eval_point);
    GkrRoundProof rp;
    E claim = num_eval * lambda + den_eval;
    rp.sumcheck.claimed_sum = claim;
    std::vector<E> alphas;
    const std::vector<E> pts(eval_point.rbegin(), eval_point.rend());
    gkr_dense_rounds(n0, d0, n1, d1, eq, pts, lambda, claim, ch, rp.sumcheck.polys, alphas);
    rp.sumcheck.eval = claim;
    rp.sumcheck.point.assign(alphas.rbegin(), alphas.rend());
    rp.numerator_0 = n0[0]; rp.denominator_0 = d0[0]; rp.numerator_1 = n1[0]; rp.denominator_1 = d1[0];
    return rp;
}

static inline std::vector<E> padded_column_evals(const F* data, size_t real_rows, int width, int L, const std::vector<E>& eq) {
    std::vector<E> out(width, E::zero());
    for (size_t r = 0; r < real_rows; r++)
        for (int c = 0; c < width; c++) out[c] += eq[r] * data[r * width + c];
    (void)L;
    return out;
}

// prove_logup_gkr. L = number of row variables of the (padded) traces = max_log_row_count.
static inline GkrProof gkr_prove(const std::vector<GkrChip>& chips, int L, Challenger& ch) {
    GkrProof proof;
    const int beta_seed_dim = gkr_beta_seed_dim(chips);
    proof.witness = ch.grind(GKR_GRINDING_BITS);
    const E alpha = ch.sample_ext();
    const std::vector<E> beta_seed = sample_point(ch, beta_seed_dim);
    (void)ch.sample_ext();                                   // _pv_challenge
    const std::vector<E> betas = partial_lagrange(beta_seed);
    size_t num_interactions = 0;
    for (auto& c : chips) num_interactions += c.interactions.size();
    const int niv = log2_ceil(num_interactions);
    const size_t W = (size_t)1 << niv;

    // C_L: per-row fractions, dense, padding = (0, 1)
    FracTable cur;
    cur.rows = (size_t)1 << L; cur.width = W;
    cur.n.assign(cur.rows * W, E::zero());
    cur.d.assign(cur.rows * W, E::one());
    {
        size_t off = 0;
        for (auto& c : chips) {
            for (size_t r = 0; r < c.real_rows; r++)
                for (size_t j = 0; j < c.interactions.size(); j++) {
                    F m; E d;
                    interaction_vals(c.interactions[j], c.prep ? c.prep + r * c.prep_width : nullptr, c.main + r * c.main_width, alpha, betas, &m, &d);
                    cur.n[r * W + off + j] = E::from_base(m);
                    cur.d[r * W + off + j] = d;
                }
            off += c.interactions.size();
        }
    }
    // C_v for v = L .. 1 (combine rows 2r, 2r+1)
    std::vector<FracTable> C(L + 1);
    C[L] = std::move(cur);
    for (int v = L - 1; v >= 1; v--) {
        const FracTable& up = C[v + 1];
        FracTable t;
        t.rows = up.rows / 2; t.width = W;
        t.n.resize(t.rows * W); t.d.resize(t.rows * W);
#pragma omp parallel for schedule(static)
        for (size_t r = 0; r < t.rows; r++)
            for (size_t i = 0; i < W; i++) {
                const E &na = up.n[(2 * r) * W + i], &da = up.d[(2 * r) * W + i], &nb = up.n[(2 * r + 1) * W + i], &db = up.d[(2 * r + 1) * W + i];
                t.n[r * W + i] = db * na + da * nb;
                t.d[r * W + i] = da * db;
            }
        C[v] = std::move(t);
    }
    // circuit output = C_1: index 2 i + r
    proof.numerator.resize(2 * W); proof.denominator.resize(2 * W);
    for (size_t i = 0; i < W; i++)
        for (size_t r = 0; r < 2; r++) { proof.numerator[2 * i + r] = C[1].n[r * W + i]; proof.denominator[2 * i + r] = C[1].d[r * W + i]; }
    ch.observe(F::from_canonical((uint32_t)proof.numerator.size()));
    for (auto& e : proof.numerator) ch.observe_ext(e);
    ch.observe(F::from_canonical((uint32_t)proof.denominator.size()));
    for (auto& e : proof.denominator) ch.observe_ext(e);
    std::vector<E> eval_point = sample_point(ch, niv + 1);
    E num_eval = eval_ext_mle_at_point(proof.numerator, eval_point), den_eval = eval_ext_mle_at_point(proof.denominator, eval_point);

    for (int v = 1; v <= L - 1; v++) {
        const FracTable& up = C[v + 1];
        const size_t R = (size_t)1 << v;
        std::vector<E> n0(W * R), d0(W * R), n1(W * R), d1(W * R);
        for (size_t i = 0; i < W; i++)
            for (size_t r = 0; r < R; r++) {
                n0[i * R + r] = up.n[(2 * r) * W + i]; d0[i * R + r] = up.d[(2 * r) * W + i];
                n1[i * R + r] = up.n[(2 * r + 1) * W + i]; d1[i * R + r] = up.d[(2 * r + 1) * W + i];
            }
        GkrRoundProof rp = gkr_round_dense(std::move(n0), std::move(d0), std::move(n1), std::move(d1), eval_point, num_eval, den_eval, ch);
        ch.observe_ext(rp.numerator_0); ch.observe_ext(rp.numerator_1);
        ch.observe_ext(rp.denominator_0); ch.observe_ext(rp.denominator_1);
        eval_point = rp.sumcheck.point;
        const E lc = ch.sample_ext();
        num_eval = rp.numerator_0 + (rp.numerator_1 - rp.numerator_0) * lc;
        den_eval = rp.denominator_0 + (rp.denominator_1 - rp.denominator_0) * lc;
        eval_point.push_back(lc);
        proof.rounds.push_back(std::move(rp));
    }
    // chip openings at the trace point
    proof.point = last_k(eval_point, L);
    const std::vector<E> eq = partial_lagrange(proof.point);
    ch.observe(F::from_canonical((uint32_t)chips.size()));
    for (auto& c : chips) {
        proof.chip_names.push_back(c.name);
        proof.main_evals.push_back(padded_column_evals(c.main, c.real_rows, c.main_width, L, eq));
        proof.has_prep.push_back(c.prep_width > 0);
        proof.prep_evals.push_back(c.prep_width > 0 ? padded_column_evals(c.prep, c.real_rows, c.prep_width, L, eq) : std::vector<E>());
        if (c.prep_width > 0) {
            ch.observe(F::from_canonical((uint32_t)c.prep_width));
            for (auto& e : proof.prep_evals.back()) ch.observe_ext(e);
        }
        ch.observe(F::from_canonical((uint32_t)c.main_width));
        for (auto& e : proof.main_evals.back()) ch.observe_ext(e);
    }
    return proof;
}

// ---- the same prover on REAL rows only (jagged-aware), for CPU timing at sizes the dense formulation cannot reach.
// This is the shape of the reference's CPU prover: every chip's layer lives over its real rows
// (/root/reference/crates/hypercube/src/logup_gkr/execution.rs:L112-L382: `generate_first_layer`, `layer_transition` on
// PaddedMle's) and the padding rows — the constant fraction (0, 1) — enter the round polynomial in closed form
// (`eq_correction_term`, logup_poly.rs:L521-L530). The dense prover above wastes a factor 2^L / rows on a scaled-down shard
// (1000x at 1 / 4096 of a core shard) and made the CPU baseline meaningless. Same field elements:
// tests/test_oracle_gkr.py checks the two provers byte for byte. The once-per-layer interaction-variable rounds run on the
// dense 2^niv-entry tables through gkr_dense_rounds.
static inline GkrProof gkr_prove_sparse(const std::vector<GkrChip>& chips, int L, Challenger& ch) {
    GkrProof proof;
    const int beta_seed_dim = gkr_beta_seed_dim(chips);
    proof.witness = ch.grind(GKR_GRINDING_BITS);
    const E alpha = ch.sample_ext();
    const std::vector<E> beta_seed = sample_point(ch, beta_seed_dim);
    (void)ch.sample_ext();                                   // _pv_challenge
    const std::vector<E> betas = partial_lagrange(beta_seed);
    struct Ref { const GkrChip* chip; const GkrInteraction* in; };
    std::vector<Ref> ints;
    for (auto& c : chips) for (auto& i : c.interactions) ints.push_back(Ref{&c, &i});
    const size_t K = ints.size();
    const int niv = log2_ceil(K);
    const size_t W = (size_t)1 << niv;
    const E one = E::one(), zero = E::zero();
    auto rows_at = [&](size_t h, int l) -> size_t { return (h + (((size_t)1 << (L - l)) - 1)) >> (L - l); };

    // levels L .. 1 over real rows: lv[l][i] = (N, D) vectors of interaction i
    std::vector<std::vector<std::vector<E>>> lvN(L + 1, std::vector<std::vector<E>>(K)), lvD(L + 1, std::vector<std::vector<E>>(K));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < K; i++) {
        const GkrChip& c = *ints[i].chip;
        std::vector<E>& n = lvN[L][i];
        std::vector<E>& d = lvD[L][i];
        n.resize(c.real_rows); d.resize(c.real_rows);
        for (size_t r = 0; r < c.real_rows; r++) {
            F m;
            interaction_vals(*ints[i].in, c.prep ? c.prep + r * c.prep_width : nullptr, c.main + r * c.main_width, alpha, betas, &m, &d[r]);
            n[r] = E::from_base(m);
        }
        for (int l = L - 1; l >= 1; l--) {
            const std::vector<E>&un = lvN[l + 1][i], &ud = lvD[l + 1][i];
            const size_t ro = (un.size() + 1) / 2;
            lvN[l][i].resize(ro); lvD[l][i].resize(ro);
            for (size_t r = 0; r < ro; r++) {
                if (2 * r + 1 < un.size()) {
                    lvN[l][i][r] = ud[2 * r + 1] * un[2 * r] + ud[2 * r] * un[2 * r + 1];
                    lvD[l][i][r] = ud[2 * r] * ud[2 * r + 1];
                } else { lvN[l][i][r] = un[2 * r]; lvD[l][i][r] = ud[2 * r]; }          // partner is a padding row (0, 1)
            }
        }
    }
    proof.numerator.assign(2 * W, zero); proof.denominator.assign(2 * W, one);
    for (size_t i = 0; i < K; i++)
        for (size_t r = 0; r < lvN[1][i].size(); r++) { proof.numerator[2 * i + r] = lvN[1][i][r]; proof.denominator[2 * i + r] = lvD[1][i][r]; }
    ch.observe(F::from_canonical((uint32_t)proof.numerator.size()));
    for (auto& e : proof.numerator) ch.observe_ext(e);
    ch.observe(F::from_canonical((uint32_t)proof.denominator.size()));
    for (auto& e : proof.denominator) ch.observe_ext(e);
    std::vector<E> eval_point = sample_point(ch, niv + 1);
    E num_eval = eval_ext_mle_at_point(proof.numerator, eval_point), den_eval = eval_ext_mle_at_point(proof.denominator, eval_point);
    const E inv2 = einv(E::from_base(F::two())), inv8 = einv(E::from_base(F::from_canonical(8))), four = E::from_base(F::from_canonical(4));
    (void)rows_at;

    for (int v = 1; v <= L - 1; v++) {
        const E lambda = ch.sample_ext();
        GkrRoundProof rp;
        E claim = num_eval * lambda + den_eval;
        rp.sumcheck.claimed_sum = claim;
        const std::vector<E> int_point(eval_point.begin(), eval_point.begin() + niv), row_point(eval_point.begin() + niv, eval_point.end());
        const std::vector<E> eq_int = partial_lagrange(int_point);
        // the layer's four tables over real rows: row r = entries 2r, 2r + 1 of level v + 1
        std::vector<std::vector<E>> t[4];
        for (int w = 0; w < 4; w++) t[w].resize(K);
        for (size_t i = 0; i < K; i++) {
            const std::vector<E>&un = lvN[v + 1][i], &ud = lvD[v + 1][i];
            const size_t live = (un.size() + 1) / 2;
            for (int w = 0; w < 4; w++) t[w][i].resize(live);
            for (size_t r = 0; r < live; r++) {
                t[0][i][r] = un[2 * r]; t[1][i][r] = ud[2 * r];
                const bool has = 2 * r + 1 < un.size();
                t[2][i][r] = has ? un[2 * r + 1] : zero; t[3][i][r] = has ? ud[2 * r + 1] : one;
            }
        }
        std::vector<E> alphas;
        E PA = one;                                          // eq factor of the row variables bound so far
        for (int j = 0; j < v; j++) {
            const int tt = v - j;                            // remaining row variables
            const std::vector<E> T = partial_lagrange(std::vector<E>(row_point.begin(), row_point.begin() + tt));
            E S0 = zero, Sh = zero, Seq = zero;
#pragma omp parallel
            {
                E l0 = zero, lh = zero, le = zero;
#pragma omp for schedule(dynamic, 4) nowait
                for (size_t i = 0; i < K; i++) {
                    const size_t live = t[0][i].size(), pairs = (live + 1) / 2;
                    E a0 = zero, ah = zero, ae = zero;
                    for (size_t k = 0; k < pairs; k++) {
                        const size_t a = 2 * k, b = 2 * k + 1;
                        const bool hb = b < live;
                        const E n0a = t[0][i][a], d0a = t[1][i][a], n1a = t[2][i][a], d1a = t[3][i][a];
                        const E n0b = hb ? t[0][i][b] : zero, d0b = hb ? t[1][i][b] : one, n1b = hb ? t[2][i][b] : zero, d1b = hb ? t[3][i][b] : one;
                        a0 += T[a] * (lambda * (n0a * d1a + n1a * d0a) + d0a * d1a);
                        const E sn0 = n0a + n0b, sn1 = n1a + n1b, sd0 = d0a + d0b, sd1 = d1a + d1b, ts = T[a] + T[b];
                        ah += ts * (lambda * (sn0 * sd1 + sn1 * sd0) + sd0 * sd1);
                        ae += ts;
                    }
                    l0 += eq_int[i] * a0; lh += eq_int[i] * ah; le += eq_int[i] * ae;
                }
#pragma omp critical
                { S0 += l0; Sh += lh; Seq += le; }
            }
            const E pt = row_point[tt - 1];
            const E corr = one - Seq;                        // eq mass of the all-padding pairs (F = 1 there)
            const E p0 = PA * (S0 + corr * (one - pt));
            const E ph = PA * (Sh + corr * four) * inv8;
            const E b_const = (one - pt) * einv(one - (pt + pt));
            UniPoly uni = interpolate_univariate({zero, one, inv2, b_const}, {p0, claim - p0, ph, zero});
            for (auto& c : uni) ch.observe_ext(c);
            rp.sumcheck.polys.push_back(uni);
            const E ar = ch.sample_ext();
            alphas.push_back(ar);
            claim = uni_eval(uni, ar);
            PA = PA * (pt * ar + (one - pt) * (one - ar));
#pragma omp parallel for schedule(dynamic, 4)
            for (size_t i = 0; i < K; i++) {
                const size_t live = t[0][i].size(), ro = (live + 1) / 2;
                for (int w = 0; w < 4; w++) {
                    std::vector<E>& x = t[w][i];
                    const E pad = (w & 1) ? one : zero;
                    for (size_t r = 0; r < ro; r++) {
                        const E lo = x[2 * r], hi = 2 * r + 1 < live ? x[2 * r + 1] : pad;
                        x[r] = lo + ar * (hi - lo);
                    }
                    x.resize(ro);
                }
            }
        }
        // interaction-variable rounds on the dense 2^niv tables (padding interactions = (0, 1))
        std::vector<E> n0(W, zero), d0(W, one), n1(W, zero), d1(W, one), eq(W);
        for (size_t i = 0; i < K; i++)
            if (!t[0][i].empty()) { n0[i] = t[0][i][0]; d0[i] = t[1][i][0]; n1[i] = t[2][i][0]; d1[i] = t[3][i][0]; }
        for (size_t i = 0; i < W; i++) eq[i] = eq_int[i] * PA;
        const std::vector<E> pts(int_point.rbegin(), int_point.rend());
        gkr_dense_rounds(n0, d0, n1, d1, eq, pts, lambda, claim, ch, rp.sumcheck.polys, alphas);
        rp.sumcheck.eval = claim;
        rp.sumcheck.point.assign(alphas.rbegin(), alphas.rend());
        rp.numerator_0 = n0[0]; rp.denominator_0 = d0[0]; rp.numerator_1 = n1[0]; rp.denominator_1 = d1[0];
        ch.observe_ext(rp.numerator_0); ch.observe_ext(rp.numerator_1);
        ch.observe_ext(rp.denominator_0); ch.observe_ext(rp.denominator_1);
        eval_point = rp.sumcheck.point;
        const E lc = ch.sample_ext();
        num_eval = rp.numerator_0 + (rp.numerator_1 - rp.numerator_0) * lc;
        den_eval = rp.denominator_0 + (rp.denominator_1 - rp.denominator_0) * lc;
        eval_point.push_back(lc);
        proof.rounds.push_back(std::move(rp));
    }
    proof.point = last_k(eval_point, L);
    const std::vector<E> eq = partial_lagrange(proof.point);
    ch.observe(F::from_canonical((uint32_t)chips.size()));
    proof.main_evals.resize(chips.size()); proof.prep_evals.resize(chips.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < chips.size(); k++) {
        const GkrChip& c = chips[k];
        proof.main_evals[k] = padded_column_evals(c.main, c.real_rows, c.main_width, L, eq);
        proof.prep_evals[k] = c.prep_width > 0 ? padded_column_evals(c.prep, c.real_rows, c.prep_width, L, eq) : std::vector<E>();
    }
    for (size_t k = 0; k < chips.size(); k++) {
        const GkrChip& c = chips[k];
        proof.chip_names.push_back(c.name);
        proof.has_prep.push_back(c.prep_width > 0);
        if (c.prep_width > 0) {
            ch.observe(F::from_canonical((uint32_t)c.prep_width));
            for (auto& e : proof.prep_evals[k]) ch.observe_ext(e);
        }
        ch.observe(F::from_canonical((uint32_t)c.main_width));
        for (auto& e : proof.main_evals[k]) ch.observe_ext(e);
    }
    return proof;
}

// verify_logup_gkr. heights[k] = real rows of chip k. check_interactions = false skips the cumulative-sum
// and the final interaction check (used on the reference's real proof, whose chips are not available).
// `pvp` / `publics`: the machine's eval_public_values and the shard's public values; null = a machine without one (cumulative
// sum zero). Codes: 4 = cumulative sum mismatch, 9 = the public values violate their constraints.
static inline int gkr_verify(const std::vector<GkrChip>& chips, const std::vector<size_t>& heights, int L, const GkrProof& proof,
                             bool check_interactions, int beta_seed_dim_override, Challenger& ch, const PvProgram* pvp = nullptr,
                             const std::vector<F>* publics = nullptr) {
    int beta_seed_dim = beta_seed_dim_override >= 0 ? beta_seed_dim_override : gkr_beta_seed_dim(chips);
    if (beta_seed_dim_override < 0 && pvp) {     // max(max_interaction_arity, max_interaction_kinds_values) (verifier.rs:L112-L126)
        size_t arity = pvp->max_kind_arity;
        for (auto& c : chips) for (auto& i : c.interactions) arity = std::max(arity, i.values.size() + 1);
        beta_seed_dim = log2_ceil(arity);
    }
    if (!ch.check_witness(GKR_GRINDING_BITS, proof.witness)) return 1;
    const E alpha = ch.sample_ext();
    const std::vector<E> beta_seed = sample_point(ch, beta_seed_dim);
    const E pv_challenge = ch.sample_ext();
    E cumulative_sum = E::zero();
    if (pvp && check_interactions) {
        if (!publics) return 9;
        E digest;
        if (!verify_public_values(*pvp, pv_challenge, alpha, partial_lagrange(beta_seed), *publics, &digest)) return 9;
        cumulative_sum = -digest;
    }
    size_t num_interactions = 0;
    for (auto& c : chips) num_interactions += c.interactions.size();
    const int niv = check_interactions ? log2_ceil(num_interactions) : log2_ceil(proof.numerator.size()) - 1;
    const size_t expected = (size_t)1 << (niv + 1);
    if (proof.numerator.size() != expected || proof.denominator.size() != expected) return 2;
    ch.observe(F::from_canonical((uint32_t)expected));
    for (auto& e : proof.numerator) ch.observe_ext(e);
    ch.observe(F::from_canonical((uint32_t)expected));
    for (auto& e : proof.denominator) ch.observe_ext(e);
    for (auto& d : proof.denominator) if (d == E::zero()) return 3;
    if (check_interactions) {
        E sum = E::zero();
        for (size_t k = 0; k < expected; k++) sum += proof.numerator[k] * einv(proof.denominator[k]);
        if (sum != cumulative_sum) return 4;         // output_cumulative_sum != cumulative_sum (verifier.rs:L168-L181)
    }
    std::vector<E> eval_point = sample_point(ch, niv + 1);
    E num_eval = eval_ext_mle_at_point(proof.numerator, eval_point), den_eval = eval_ext_mle_at_point(proof.denominator, eval_point);
    if ((int)proof.rounds.size() + 1 != L) return 2;
    for (size_t i = 0; i < proof.rounds.size(); i++) {
        const GkrRoundProof& rp = proof.rounds[i];
        const E lambda = ch.sample_ext();
        if (rp.sumcheck.claimed_sum != num_eval * lambda + den_eval) return 5;
        if (int rc = partially_verify_sumcheck(rp.sumcheck, ch, i + niv + 1, 3)) return 10 + rc;
        const E eq_eval = full_lagrange_eval(rp.sumcheck.point, eval_point);
        const E nse = rp.numerator_0 * rp.denominator_1 + rp.numerator_1 * rp.denominator_0;
        const E dse = rp.denominator_0 * rp.denominator_1;
        if (rp.sumcheck.eval != eq_eval * (nse * lambda + dse)) return 6;
        ch.observe_ext(rp.numerator_0); ch.observe_ext(rp.numerator_1);
        ch.observe_ext(rp.denominator_0); ch.observe_ext(rp.denominator_1);
        eval_point = rp.sumcheck.point;
        const E lc = ch.sample_ext();
        eval_point.push_back(lc);
        num_eval = rp.numerator_0 + (rp.numerator_1 - rp.numerator_0) * lc;
        den_eval = rp.denominator_0 + (rp.denominator_1 - rp.denominator_0) * lc;
    }
    const std::vector<E> interaction_point(eval_point.begin(), eval_point.begin() + niv), trace_point(eval_point.begin() + niv, eval_point.end());
    if ((int)trace_point.size() != L) return 7;
    if (proof.point != trace_point) return 8;
    const size_t n_chips = proof.chip_names.size();
    ch.observe(F::from_canonical((uint32_t)n_chips));
    for (size_t k = 0; k < n_chips; k++) {
        if (proof.has_prep[k]) {
            ch.observe(F::from_canonical((uint32_t)proof.prep_evals[k].size()));
            for (auto& e : proof.prep_evals[k]) ch.observe_ext(e);
        }
        ch.observe(F::from_canonical((uint32_t)proof.main_evals[k].size()));
        for (auto& e : proof.main_evals[k]) ch.observe_ext(e);
    }
    if (!check_interactions) return 0;
    if (n_chips != chips.size()) return 2;
    const std::vector<E> betas = partial_lagrange(beta_seed);
    std::vector<E> nums, dens;
    std::vector<E> point_ext = trace_point;
    point_ext.insert(point_ext.begin(), E::zero());
    for (size_t k = 0; k < n_chips; k++) {
        const GkrChip& c = chips[k];
        if (proof.chip_names[k] != c.name || (int)proof.main_evals[k].size() != c.main_width ||
            (proof.has_prep[k] ? (int)proof.prep_evals[k].size() : 0) != c.prep_width)
            return 2;
        std::vector<F> thr(L + 1);
        for (int b = 0; b <= L; b++) thr[b] = F::from_canonical((uint32_t)((heights[k] >> (L - b)) & 1));
        const E geq = full_geq(thr, point_ext);
        const std::vector<E> zm(c.main_width, E::zero()), zp(c.prep_width, E::zero());
        for (auto& in : c.interactions) {
            E rn, rd, pn, pd;
            interaction_eval_ext(in, proof.prep_evals[k].data(), proof.main_evals[k].data(), alpha, betas, &rn, &rd);
            interaction_eval_ext(in, zp.data(), zm.data(), alpha, betas, &pn, &pd);
            E ne = rn - pn * geq;
            const E de = rd + (E::one() - pd) * geq;
            if (!in.is_send) ne = -ne;
            nums.push_back(ne);
            dens.push_back(de);
        }
    }
    nums.resize((size_t)1 << niv, E::zero());
    dens.resize((size_t)1 << niv, E::one());
    if (num_eval != eval_ext_mle_at_point(nums, interaction_point)) return 20;
    if (den_eval != eval_ext_mle_at_point(dens, interaction_point)) return 21;
    return 0;
}

// ------------------------------------------------------------------ bincode(LogupGkrProof)
static inline std::vector<uint8_t> serialize_gkr_proof(const GkrProof& p) {
    ByteWriter w;
    for (const std::vector<E>* v : {&p.numerator, &p.denominator}) {
        w.u64(v->size());
        for (auto& e : *v) w.e(e);
        w.u64(2); w.u64(v->size()); w.u64(1);
    }
    w.u64(p.rounds.size());
    for (auto& r : p.rounds) {
        w.e(r.numerator_0); w.e(r.numerator_1); w.e(r.denominator_0); w.e(r.denominator_1);
        write_sumcheck(w, r.sumcheck);
    }
    w.u64(p.point.size());
    for (auto& e : p.point) w.e(e);
    w.u64(p.chip_names.size());
    for (size_t k = 0; k < p.chip_names.size(); k++) {
        w.u64(p.chip_names[k].size());
        for (char c : p.chip_names[k]) w.b.push_back((uint8_t)c);
        w.u64(p.main_evals[k].size());
        for (auto& e : p.main_evals[k]) w.e(e);
        w.u64(1); w.u64(p.main_evals[k].size());
        w.b.push_back(p.has_prep[k] ? 1 : 0);
        if (p.has_prep[k]) {
            w.u64(p.prep_evals[k].size());
            for (auto& e : p.prep_evals[k]) w.e(e);
            w.u64(1); w.u64(p.prep_evals[k].size());
        }
    }
    w.f(p.witness);
    return w.b;
}

static inline GkrProof deserialize_gkr_proof(const uint8_t* buf, size_t len) {
    ByteReader r{buf, len};
    GkrProof p;
    auto read_vec = [&](std::vector<E>& v) {
        size_t n = r.u64();
        if (n > len) throw std::runtime_error("bad length");
        v.resize(n);
        for (auto& e : v) e = r.e();
    };
    for (std::vector<E>* v : {&p.numerator, &p.denominator}) {
        read_vec(*v);
        if (r.u64() != 2 || r.u64() != v->size() || r.u64() != 1) throw std::runtime_error("bad Mle shape");
    }
    size_t n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    p.rounds.resize(n);
    for (auto& rp : p.rounds) {
        rp.numerator_0 = r.e(); rp.numerator_1 = r.e(); rp.denominator_0 = r.e(); rp.denominator_1 = r.e();
        rp.sumcheck = read_sumcheck(r);
    }
    read_vec(p.point);
    n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    for (size_t k = 0; k < n; k++) {
        size_t sl = r.u64();
        if (sl > 256) throw std::runtime_error("bad name");
        r.need(sl);
        p.chip_names.emplace_back((const char*)r.p + r.o, sl);
        r.o += sl;
        std::vector<E> m;
        read_vec(m);
        if (r.u64() != 1 || r.u64() != m.size()) throw std::runtime_error("bad MleEval shape");
        p.main_evals.push_back(m);
        r.need(1);
        const bool hp = r.p[r.o++] != 0;
        p.has_prep.push_back(hp);
        std::vector<E> pe;
        if (hp) {
            read_vec(pe);
            if (r.u64() != 1 || r.u64() != pe.size()) throw std::runtime_error("bad MleEval shape");
        }
        p.prep_evals.push_back(pe);
    }
    p.witness = r.f();
    if (r.o != len) throw std::runtime_error("trailing bytes");
    return p;
}

}  // namespace orc
