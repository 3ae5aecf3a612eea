// oracle/kb_shard.hpp — TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
//
// The whole shard proof: `ShardProver::prove_shard_with_data`
// (/root/reference/crates/hypercube/src/prover/shard.rs:L650-L792) and `ShardVerifier::verify_shard`
// (/root/reference/crates/hypercube/src/verifier/shard.rs:L437-L742) assembled from the pieces restated in
// kb_pcs.hpp / kb_zerocheck.hpp / kb_jagged.hpp / kb_gkr.hpp, plus bincode(ShardProof)
// (/root/reference/crates/hypercube/src/verifier/proof.rs:L47-L94).
// Pinned by reference data: `shard_verify` in transcript-only mode parses the reference's REAL ShardProof
// (tests/golden: its own bytes, BaseFold queries trimmed to 12) and accepts everything that does not need the
// recursion machine's chip definitions, ending in the exact transcript state; see tests/test_oracle_golden.py.
#pragma once
#include <chrono>
#include "kb_gkr.hpp"

namespace orc {

struct ShardChip {
    std::string name;
    ZcAir air;                                   // widths + constraint program
    std::vector<GkrInteraction> interactions;
    const F* main = nullptr;                     // [real_rows][air.main_width] row-major
    const F* prep = nullptr;
    size_t real_rows = 0;
};

struct ShardProof {
    std::vector<F> public_values;
    Digest main_commitment;
    GkrProof gkr;
    ZcProof zerocheck;                           // chip_evals[k] = preprocessed then main openings at the zerocheck point
    std::vector<std::string> names;              // opened_values keys
    std::vector<size_t> prep_widths;             // split of chip_evals[k]
    std::vector<size_t> heights;                 // from the degree bit strings
    size_t degree_bits = 0;
    JaggedProof evaluation;
};

static inline std::vector<uint8_t> serialize_shard_proof(const ShardProof& p) {
    ByteWriter w;
    w.u64(p.public_values.size());
    for (auto& x : p.public_values) w.f(x);
    w.d(p.main_commitment);
    { auto b = serialize_gkr_proof(p.gkr); w.b.insert(w.b.end(), b.begin(), b.end()); }
    {
        SumcheckProof sc{p.zerocheck.univariate_polys, p.zerocheck.claimed_sum, p.zerocheck.point, p.zerocheck.eval};
        write_sumcheck(w, sc);
    }
    w.u64(p.names.size());
    for (size_t k = 0; k < p.names.size(); k++) {
        w.u64(p.names[k].size());
        for (char c : p.names[k]) w.b.push_back((uint8_t)c);
        const auto& ev = p.zerocheck.chip_evals[k];
        w.u64(p.prep_widths[k]);
        for (size_t c = 0; c < p.prep_widths[k]; c++) w.e(ev[c]);
        w.u64(ev.size() - p.prep_widths[k]);
        for (size_t c = p.prep_widths[k]; c < ev.size(); c++) w.e(ev[c]);
        w.u64(p.degree_bits);
        for (size_t b = 0; b < p.degree_bits; b++) w.f(F::from_canonical((uint32_t)((p.heights[k] >> (p.degree_bits - 1 - b)) & 1)));
    }
    { auto b = serialize_jagged_proof(p.evaluation); w.b.insert(w.b.end(), b.begin(), b.end()); }
    return w.b;
}

// the sub-proof (de)serialisers take whole buffers; find their extents by parsing
static inline ShardProof deserialize_shard_proof(const uint8_t* buf, size_t len) {
    ByteReader r{buf, len};
    ShardProof p;
    size_t n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    p.public_values.resize(n);
    for (auto& x : p.public_values) x = r.f();
    p.main_commitment = r.d();
    {   // LogupGkrProof: walk it to find its end
        const size_t start = r.o;
        for (int k = 0; k < 2; k++) { size_t m = r.u64(); r.need(m * 16); r.o += m * 16; r.need(24); r.o += 24; }
        size_t nr = r.u64();
        for (size_t k = 0; k < nr; k++) { r.need(64); r.o += 64; (void)read_sumcheck(r); }
        size_t np = r.u64(); r.need(np * 16); r.o += np * 16;
        size_t nc = r.u64();
        for (size_t k = 0; k < nc; k++) {
            size_t sl = r.u64(); r.need(sl); r.o += sl;
            size_t m = r.u64(); r.need(m * 16 + 16); r.o += m * 16 + 16;
            r.need(1);
            if (r.p[r.o++]) { size_t q = r.u64(); r.need(q * 16 + 16); r.o += q * 16 + 16; }
        }
        r.need(4); r.o += 4;
        p.gkr = deserialize_gkr_proof(buf + start, r.o - start);
    }
    {
        SumcheckProof sc = read_sumcheck(r);
        p.zerocheck.univariate_polys = sc.polys; p.zerocheck.claimed_sum = sc.claimed_sum;
        p.zerocheck.point = sc.point; p.zerocheck.eval = sc.eval;
    }
    n = r.u64();
    if (n > len) throw std::runtime_error("bad length");
    for (size_t k = 0; k < n; k++) {
        size_t sl = r.u64();
        if (sl > 256) throw std::runtime_error("bad name");
        r.need(sl);
        p.names.emplace_back((const char*)r.p + r.o, sl);
        r.o += sl;
        std::vector<E> ev;
        size_t np = r.u64();
        if (np > len) throw std::runtime_error("bad length");
        for (size_t c = 0; c < np; c++) ev.push_back(r.e());
        size_t nm = r.u64();
        if (nm > len) throw std::runtime_error("bad length");
        for (size_t c = 0; c < nm; c++) ev.push_back(r.e());
        p.prep_widths.push_back(np);
        p.zerocheck.chip_evals.push_back(ev);
        size_t nb = r.u64();
        if (nb > 64) throw std::runtime_error("bad degree");
        if (k == 0) p.degree_bits = nb; else if (nb != p.degree_bits) throw std::runtime_error("degree lengths differ");
        size_t h = 0;
        for (size_t b = 0; b < nb; b++) { const uint32_t bit = r.f().canonical(); if (bit > 1) throw std::runtime_error("degree not boolean"); h = 2 * h + bit; }
        p.heights.push_back(h);
    }
    p.evaluation = deserialize_jagged_proof(buf + r.o, len - r.o);
    return p;
}

struct ShardParams { int max_log_row_count, log_stacking_height; size_t batch_size; FriConfig fri; };

static inline std::vector<GkrChip> gkr_chips_of(const std::vector<ShardChip>& chips) {
    std::vector<GkrChip> out;
    for (auto& c : chips) {
        GkrChip g;
        g.name = c.name; g.interactions = c.interactions; g.main_width = c.air.main_width; g.prep_width = c.air.prep_width;
        g.main = c.main; g.prep = c.prep; g.real_rows = c.real_rows;
        out.push_back(std::move(g));
    }
    return out;
}

// prove_shard_with_data. `prep_round`: the preprocessed commitment round of the proving key (JaggedRoundData of
// the chips' preprocessed traces, committed at setup). The challenger has already absorbed the verifying key.
// wall seconds of the last shard_prove on this thread: commit, LogUp-GKR, zerocheck, jagged evaluation proof (CPU baseline)
static thread_local double g_stage_seconds[4] = {0, 0, 0, 0};
// 1: LogUp-GKR over real rows only (gkr_prove_sparse: the reference's CPU shape); 0: the dense, independent formulation
static int g_gkr_sparse = 0;

static inline ShardProof shard_prove(const std::vector<ShardChip>& chips, const std::vector<F>& publics,
                                     const JaggedRoundData& prep_round, const ShardParams& sp, Challenger& ch) {
    const int L = sp.max_log_row_count;
    auto t_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = t_now();
    ShardProof proof;
    proof.public_values = publics;
    for (auto& x : publics) ch.observe(x);
    std::vector<TensorRef> tables;
    for (auto& c : chips) tables.push_back(TensorRef{c.main, c.real_rows, c.air.main_width});
    JaggedRoundData main_round;
    proof.main_commitment = jagged_commit(tables, L, sp.log_stacking_height, sp.batch_size, sp.fri, &main_round);
    ch.observe_digest(proof.main_commitment);
    ch.observe(F::from_canonical((uint32_t)chips.size()));
    for (auto& c : chips) {
        ch.observe(F::from_canonical((uint32_t)c.real_rows));
        ch.observe(F::from_canonical((uint32_t)c.name.size()));
        for (char b : c.name) ch.observe(F::from_canonical((uint8_t)b));
    }
    g_stage_seconds[0] = t_now() - t0; t0 = t_now();
    proof.gkr = g_gkr_sparse ? gkr_prove_sparse(gkr_chips_of(chips), L, ch) : gkr_prove(gkr_chips_of(chips), L, ch);
    g_stage_seconds[1] = t_now() - t0; t0 = t_now();
    const E batching = ch.sample_ext(), gkr_batch = ch.sample_ext();
    std::vector<ZcChipInput> zc(chips.size());
    for (size_t k = 0; k < chips.size(); k++) {
        zc[k].air = &chips[k].air; zc[k].main = chips[k].main; zc[k].prep = chips[k].prep; zc[k].real_rows = chips[k].real_rows;
        zc[k].main_opening = proof.gkr.main_evals[k];
        zc[k].prep_opening = proof.gkr.prep_evals[k];
    }
    proof.zerocheck = zerocheck_prove(zc, L, proof.gkr.point, batching, gkr_batch, publics, ch);
    g_stage_seconds[2] = t_now() - t0; t0 = t_now();
    std::vector<E> prep_claims, main_claims;
    for (size_t k = 0; k < chips.size(); k++) {
        const auto& ev = proof.zerocheck.chip_evals[k];
        const size_t pw = chips[k].air.prep_width;
        prep_claims.insert(prep_claims.end(), ev.begin(), ev.begin() + pw);
        main_claims.insert(main_claims.end(), ev.begin() + pw, ev.end());
        proof.names.push_back(chips[k].name);
        proof.prep_widths.push_back(pw);
        proof.heights.push_back(chips[k].real_rows);
    }
    proof.degree_bits = L + 1;
    proof.evaluation = jagged_prove(proof.zerocheck.point, {prep_claims, main_claims}, {prep_round, main_round}, L,
                                    sp.log_stacking_height, sp.fri, ch);
    g_stage_seconds[3] = t_now() - t0;
    return proof;
}

// verify_shard. with_chips = false: everything that does not need chip definitions (used on the reference's real
// proof; `beta_seed_dim` must then be given). Returns 0 when accepted.
// `pvp`: the machine's eval_public_values (kb_gkr.hpp PvProgram) or null for a machine without one. Code 4: public values of
// the wrong length or non-zero beyond the machine's words.
static inline int shard_verify(const std::vector<ShardChip>& chips, const Digest& preprocessed_commit, const ShardProof& proof,
                               const ShardParams& sp, bool with_chips, int beta_seed_dim, Challenger& ch, const PvProgram* pvp = nullptr) {
    const int L = sp.max_log_row_count;
    const size_t n = proof.names.size();
    if (proof.degree_bits != (size_t)L + 1) return 1;
    if (pvp && pvp->proof_max_num_pvs) {             // verifier/shard.rs:L455-L464
        if (proof.public_values.size() != (size_t)pvp->proof_max_num_pvs || proof.public_values.size() < (size_t)pvp->num_pv_elts) return 4;
        for (size_t i = pvp->num_pv_elts; i < proof.public_values.size(); i++) if (proof.public_values[i] != F::zero()) return 4;
    }
    for (auto& x : proof.public_values) ch.observe(x);
    ch.observe_digest(proof.main_commitment);
    ch.observe(F::from_canonical((uint32_t)n));
    for (size_t k = 0; k < n; k++) {
        ch.observe(F::from_canonical((uint32_t)proof.heights[k]));
        ch.observe(F::from_canonical((uint32_t)proof.names[k].size()));
        for (char b : proof.names[k]) ch.observe(F::from_canonical((uint8_t)b));
    }
    if (with_chips) {
        if (chips.size() != n) return 1;
        for (size_t k = 0; k < n; k++) {
            if (chips[k].name != proof.names[k] || proof.heights[k] > ((size_t)1 << L)) return 1;
            if (proof.prep_widths[k] != (size_t)chips[k].air.prep_width ||
                proof.zerocheck.chip_evals[k].size() != (size_t)(chips[k].air.prep_width + chips[k].air.main_width))
                return 1;
        }
    }
    if (proof.gkr.chip_names != proof.names) return 2;
    if (int rc = gkr_verify(gkr_chips_of(chips), proof.heights, L, proof.gkr, with_chips, with_chips ? -1 : beta_seed_dim, ch, pvp,
                            &proof.public_values))
        return 100 + rc;
    const E batching = ch.sample_ext(), gkr_batch = ch.sample_ext();
    if (with_chips) {
        std::vector<const ZcAir*> airs;
        for (auto& c : chips) airs.push_back(&c.air);
        if (int rc = zerocheck_verify(airs, proof.heights, proof.gkr.main_evals, proof.gkr.prep_evals, L, proof.gkr.point, batching,
                                      gkr_batch, proof.public_values, proof.zerocheck, ch))
            return 200 + rc;
    } else {
        (void)ch.sample_ext();                               // lambda
        SumcheckProof sc{proof.zerocheck.univariate_polys, proof.zerocheck.claimed_sum, proof.zerocheck.point, proof.zerocheck.eval};
        if (int rc = partially_verify_sumcheck(sc, ch, L, 4)) return 200 + rc;
    }
    ch.observe(F::from_canonical((uint32_t)n));
    std::vector<E> prep_claims, main_claims;
    for (size_t k = 0; k < n; k++) {
        const auto& ev = proof.zerocheck.chip_evals[k];
        const size_t pw = proof.prep_widths[k];
        ch.observe(F::from_canonical((uint32_t)pw));
        for (size_t c = 0; c < pw; c++) ch.observe_ext(ev[c]);
        ch.observe(F::from_canonical((uint32_t)(ev.size() - pw)));
        for (size_t c = pw; c < ev.size(); c++) ch.observe_ext(ev[c]);
        prep_claims.insert(prep_claims.end(), ev.begin(), ev.begin() + pw);
        main_claims.insert(main_claims.end(), ev.begin() + pw, ev.end());
    }
    if (int rc = jagged_verify({preprocessed_commit, proof.main_commitment}, proof.zerocheck.point, {prep_claims, main_claims},
                               proof.evaluation, L, sp.log_stacking_height, sp.fri, ch))
        return 300 + rc;
    // row / column counts of the jagged proof against the opened values (shard.rs:L669-L740)
    const auto& rcs = proof.evaluation.row_counts_and_column_counts;
    if (rcs.size() != 2) return 3;
    std::vector<size_t> prep_rows, main_rows, prep_cols, main_cols;
    for (size_t k = 0; k < n; k++) {
        if (proof.prep_widths[k] > 0) { prep_rows.push_back(proof.heights[k]); prep_cols.push_back(proof.prep_widths[k]); }
        main_rows.push_back(proof.heights[k]);
        main_cols.push_back(proof.zerocheck.chip_evals[k].size() - proof.prep_widths[k]);
    }
    const std::vector<size_t>* want_rows[2] = {&prep_rows, &main_rows};
    const std::vector<size_t>* want_cols[2] = {&prep_cols, &main_cols};
    for (int r = 0; r < 2; r++) {
        if (rcs[r].size() != want_rows[r]->size() + 2) return 3;
        for (size_t t = 0; t < want_rows[r]->size(); t++)
            if (rcs[r][t].first != (*want_rows[r])[t] || rcs[r][t].second != (*want_cols[r])[t]) return 3;
    }
    return 0;
}

}  // namespace orc
