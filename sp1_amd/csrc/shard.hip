// sp1_amd/csrc/shard.hip — one core shard proof, start to finish, on the device:
//
//   sp1hip_prove_shard   `ShardProver::prove_shard_with_data`   /root/reference/crates/hypercube/src/prover/shard.rs:L650-L792
//
// the method behind `AirProver::prove_shard_with_pk` (shard.rs:L45-L109), i.e. the seam at which a proving backend
// plugs into SP1. It strings the stage entry points of this library together in the reference's transcript order
//   observe public values -> commit_traces (sp1hip_jagged_commit) -> observe commitment, chip count, heights, names
//   -> LogUp-GKR (sp1hip_logup_gkr_prove) -> sample the two batching challenges -> zerocheck (sp1hip_zerocheck_prove)
//   -> jagged evaluation proof at the zerocheck point (sp1hip_jagged_prove)
// and emits bincode(ShardProof) (/root/reference/crates/hypercube/src/verifier/proof.rs:L47-L94): public_values,
// main_commitment, logup_gkr_proof, zerocheck_proof, opened_values, evaluation_proof.
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "device_ctx.hpp"
#include "stacked_data.hpp"

namespace sp1hip {
void challenger_observe(sp1hip_challenger_t* ch, uint32_t x);
kb::Ext challenger_sample_ext(sp1hip_challenger_t* ch);
void challenger_restore(sp1hip_challenger_t* dst, const sp1hip_challenger_t* src);
size_t jagged_proof_size(int lsh, const std::vector<uint32_t>& round_widths, const std::vector<size_t>& tables_per_round,
                         uint64_t total_area, sp1hip_fri_config_t config);

namespace {

struct Reader {                                   // walks our own bincode blobs
    const uint8_t* p;
    size_t n, o = 0;
    uint64_t u64() { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[o + i] << (8 * i); o += 8; return v; }
    uint32_t canonical() { uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[o + i] << (8 * i); o += 4; return v; }
    kb::Ext ext() { kb::Ext e; for (int k = 0; k < 4; k++) e.c[k] = kb::to_monty(canonical()); return e; }
    void skip(size_t k) { o += k; }
};

void put_u64(std::vector<uint8_t>& b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
void put_felt(std::vector<uint8_t>& b, uint32_t monty) { const uint32_t v = kb::from_monty(monty); for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }

}  // namespace
}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_prove_shard(const sp1hip_shard_chip_t* chips, int n_chips, const uint32_t* h_publics, int n_publics,
                       sp1hip_stacked_data_t* preprocessed, sp1hip_shard_params_t params, sp1hip_challenger_t* challenger,
                       uint8_t* h_proof, size_t* proof_len, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(chips && n_chips > 0 && preprocessed && challenger && proof_len, "bad argument");
    SP1HIP_REQUIRE(h_publics || n_publics == 0, "null public values");
    ActiveProver active;                                     // (counted for the zerocheck's fork-stream decision: common.hpp)
    const auto sh_entry = std::chrono::steady_clock::now();
    uint64_t miss_n0 = 0, miss_b0 = 0;
    arena_miss_stats(&miss_n0, &miss_b0);
    const int L = params.max_log_row_count, lsh = params.log_stacking_height;
    SP1HIP_REQUIRE(preprocessed->jagged && preprocessed->max_log_row_count == L && preprocessed->log_stacking_height == lsh,
                   "the preprocessed round was committed with different parameters");
    hipStream_t s = S(stream);

    // ---- per-stage chip descriptors
    std::vector<sp1hip_gkr_chip_t> gk(n_chips);
    std::vector<sp1hip_zc_chip_t> zc(n_chips);
    std::vector<sp1hip_table_t> tables(n_chips);
    uint64_t main_area = 0;
    size_t total_w = 0, prep_cols = 0, opened_bytes = 8;
    for (int c = 0; c < n_chips; c++) {
        const sp1hip_shard_chip_t& ci = chips[c];
        SP1HIP_REQUIRE(ci.name && ci.program && ci.interactions, "null chip field");
        SP1HIP_REQUIRE(ci.main_width > 0, "a chip needs at least one main column");
        gk[c] = sp1hip_gkr_chip_t{ci.name, ci.interactions, ci.n_words, ci.main_width, ci.prep_width, ci.d_main, ci.d_prep, ci.real_rows};
        zc[c] = sp1hip_zc_chip_t{ci.program, ci.n_instr, ci.main_width, ci.prep_width, ci.num_constraints, ci.d_main, ci.d_prep, ci.real_rows};
        tables[c] = sp1hip_table_t{ci.d_main, ci.real_rows, ci.main_width};
        main_area += ci.real_rows * (uint64_t)ci.main_width;
        total_w += ci.main_width + ci.prep_width;
        prep_cols += ci.prep_width;
        opened_bytes += 8 + strlen(ci.name) + 8 + (size_t)ci.prep_width * 16 + 8 + (size_t)ci.main_width * 16 + 8 + (size_t)(L + 1) * 4;
    }
    SP1HIP_REQUIRE(prep_cols > 0, "a shard needs at least one preprocessed column (two commitment rounds)");
    SP1HIP_REQUIRE(main_area > 0, "empty shard");
    // the proof's hand-over timeline repeats between proofs of one shape (common.hpp: WaitPlan)
    uint64_t shape_sig = 0xcbf29ce484222325ull;
    {
        auto mix = [&](uint64_t v) { shape_sig = (shape_sig ^ v) * 0x100000001b3ull; };
        mix((uint64_t)n_chips); mix((uint64_t)L); mix((uint64_t)lsh); mix((uint64_t)params.fri.log_blowup); mix((uint64_t)params.fri.num_queries);
        for (int c = 0; c < n_chips; c++) { mix(chips[c].real_rows); mix((uint64_t)chips[c].main_width << 32 | (uint32_t)chips[c].n_instr); }
    }

    // ---- exact proof size from the shapes (remembered per shape signature + interaction / program sizes: the two dry calls walk
    // every interaction and every constraint program, ~0.3 ms of host time in front of a proof's first launch)
    size_t gkr_size = 0, zc_size = 0;
    uint64_t size_sig = shape_sig;
    for (int c = 0; c < n_chips; c++) {
        size_sig = (size_sig ^ ((uint64_t)chips[c].n_words << 32 | chips[c].num_constraints)) * 0x100000001b3ull;
        size_sig = (size_sig ^ ((uint64_t)strlen(chips[c].name) << 32 | chips[c].prep_width)) * 0x100000001b3ull;
    }
    size_sig = (size_sig ^ (uint64_t)n_publics) * 0x100000001b3ull;
    static std::mutex size_mutex;
    static std::unordered_map<uint64_t, std::pair<size_t, size_t>> size_cache;
    bool have_sizes = false;
    {
        std::lock_guard<std::mutex> lk(size_mutex);
        auto it = size_cache.find(size_sig);
        if (it != size_cache.end()) { gkr_size = it->second.first; zc_size = it->second.second; have_sizes = true; }
    }
    if (!have_sizes) {
        int st = sp1hip_logup_gkr_prove(gk.data(), n_chips, L, challenger, nullptr, &gkr_size, stream);
        if (st != SP1HIP_ERROR_BUFFER_TOO_SMALL) return st == SP1HIP_SUCCESS ? SP1HIP_ERROR_RUNTIME : st;
        std::vector<sp1hip_ext_t> dummy(std::max<size_t>({total_w, (size_t)L, 1}));
        st = sp1hip_zerocheck_prove(zc.data(), n_chips, L, dummy.data(), dummy.data(), dummy[0], dummy[0], h_publics, n_publics,
                                    challenger, nullptr, &zc_size, stream);
        if (st != SP1HIP_ERROR_BUFFER_TOO_SMALL) return st == SP1HIP_SUCCESS ? SP1HIP_ERROR_RUNTIME : st;
        std::lock_guard<std::mutex> lk(size_mutex);
        if (size_cache.size() > 4096) size_cache.clear();
        size_cache[size_sig] = {gkr_size, zc_size};
    }
    const uint64_t H = (uint64_t)1 << lsh;
    const uint64_t main_padded = std::max<uint64_t>((main_area + H - 1) / H, 1) * H;
    const std::vector<uint32_t> round_widths{(uint32_t)(preprocessed->padded >> lsh), (uint32_t)(main_padded >> lsh)};
    const std::vector<size_t> tables_per_round{preprocessed->row_counts.size(), (size_t)n_chips + 2};
    const size_t jag_size = jagged_proof_size(lsh, round_widths, tables_per_round, preprocessed->padded + main_padded, params.fri);
    const size_t zc_sumcheck_size = zc_size - (8 + (size_t)n_chips * 8 + total_w * 16);
    const size_t need = 8 + (size_t)n_publics * 4 + 32 + gkr_size + zc_sumcheck_size + opened_bytes + jag_size;
    if (!h_proof || *proof_len < need) {
        *proof_len = need;
        set_error("sp1hip_prove_shard: proof buffer too small, need %zu bytes", need);
        return SP1HIP_ERROR_BUFFER_TOO_SMALL;
    }

    // SP1HIP_SHARD_TIMING=1: host wall time of the stages on stderr
    const bool sh_timing = [] { const char* e = getenv("SP1HIP_SHARD_TIMING"); return e && e[0] == '1'; }();
    std::chrono::steady_clock::time_point sh_t[6];
    sh_t[0] = std::chrono::steady_clock::now();
    // roctx ranges per stage (rocprofv3 --marker-trace): one open range at a time, closed on every exit path
    RoctxRange proof_range("sp1hip_prove_shard");
    // (with the event timers on — sp1hip_timers_enable — every stage is also a timer "stage_<name>" on the caller's stream: the
    // window the stage's kernels share, which is what bench.py prices the overlapped commit kernels against)
    struct StageMarks {
        hipStream_t s;
        bool open = false;
        int idx = -1;
        void close() { if (idx >= 0) timer_end(idx, s); idx = -1; if (open) roctx_pop(); open = false; set_stage_note(""); }
        void next(const char* name) {
            close();
            roctx_push(name);
            set_stage_note(name);
            open = true;
            if (timers_on()) { char buf[64]; snprintf(buf, sizeof buf, "stage_%s", name); idx = timer_begin(buf, s); }
        }
        ~StageMarks() { close(); }
    } marks{S(stream)};
    WaitPlan wait_plan(shape_sig);
    marks.next("commit");
    sp1hip_challenger_t* ch = nullptr;
    SP1HIP_TRY(sp1hip_challenger_clone(challenger, &ch));
    struct ChGuard { sp1hip_challenger_t* c; ~ChGuard() { sp1hip_challenger_free(c); } } guard{ch};

    // ---- transcript head + main commitment (shard.rs:L676-L698)
    for (int i = 0; i < n_publics; i++) {
        SP1HIP_REQUIRE(h_publics[i] < kb::P, "public value not reduced");
        challenger_observe(ch, h_publics[i]);
    }
    uint32_t main_commit[8];
    sp1hip_stacked_data_t* main_data = nullptr;
    SP1HIP_TRY(sp1hip_jagged_commit(tables.data(), n_chips, L, lsh, params.batch_size, params.fri.log_blowup, main_commit, &main_data, stream));
    std::unique_ptr<sp1hip_stacked_data_s> main_guard(main_data);
    for (int k = 0; k < 8; k++) challenger_observe(ch, main_commit[k]);
    challenger_observe(ch, kb::to_monty((uint32_t)n_chips));
    for (int c = 0; c < n_chips; c++) {
        challenger_observe(ch, kb::to_monty((uint32_t)chips[c].real_rows));
        const size_t nl = strlen(chips[c].name);
        challenger_observe(ch, kb::to_monty((uint32_t)nl));
        for (size_t b = 0; b < nl; b++) challenger_observe(ch, kb::to_monty((uint8_t)chips[c].name[b]));
    }

    sh_t[1] = std::chrono::steady_clock::now();
    marks.next("logup_gkr");
    // ---- LogUp-GKR
    std::vector<uint8_t> gkr_blob(gkr_size);
    size_t glen = gkr_size;
    SP1HIP_TRY(sp1hip_logup_gkr_prove(gk.data(), n_chips, L, ch, gkr_blob.data(), &glen, stream));
    // its logup_evaluations: the point and, per chip, main then preprocessed openings (the order zerocheck takes)
    std::vector<kb::Ext> zeta(L), openings;
    {
        Reader r{gkr_blob.data(), glen};
        for (int k = 0; k < 2; k++) { const uint64_t m = r.u64(); r.skip(m * 16 + 24); }
        const uint64_t nr = r.u64();
        for (uint64_t k = 0; k < nr; k++) {
            r.skip(64);
            const uint64_t np = r.u64();
            for (uint64_t q = 0; q < np; q++) { const uint64_t nc = r.u64(); r.skip(nc * 16); }
            r.skip(16);
            const uint64_t pl = r.u64();
            r.skip(pl * 16 + 16);
        }
        SP1HIP_REQUIRE(r.u64() == (uint64_t)L, "internal: GKR point dimension");
        for (auto& z : zeta) z = r.ext();
        SP1HIP_REQUIRE(r.u64() == (uint64_t)n_chips, "internal: GKR chip count");
        for (int c = 0; c < n_chips; c++) {
            r.skip(r.u64());
            const uint64_t mw = r.u64();
            for (uint64_t k = 0; k < mw; k++) openings.push_back(r.ext());
            r.skip(16);
            if (r.p[r.o++]) {
                const uint64_t pw = r.u64();
                for (uint64_t k = 0; k < pw; k++) openings.push_back(r.ext());
                r.skip(16);
            }
        }
    }
    sh_t[2] = std::chrono::steady_clock::now();
    marks.next("zerocheck");
    // ---- zerocheck
    const kb::Ext batching = challenger_sample_ext(ch), gkr_batch = challenger_sample_ext(ch);
    sp1hip_ext_t c_batching, c_gkr;
    memcpy(&c_batching, &batching, 16);
    memcpy(&c_gkr, &gkr_batch, 16);
    std::vector<uint8_t> zc_blob(zc_size);
    size_t zlen = zc_size;
    SP1HIP_TRY(sp1hip_zerocheck_prove(zc.data(), n_chips, L, reinterpret_cast<const sp1hip_ext_t*>(zeta.data()),
                                      reinterpret_cast<const sp1hip_ext_t*>(openings.data()), c_batching, c_gkr, h_publics, n_publics, ch,
                                      zc_blob.data(), &zlen, stream));
    // split: PartialSumcheckProof bytes | per-chip opened values (preprocessed then main)
    std::vector<kb::Ext> z_row(L), prep_claims, main_claims;
    std::vector<std::vector<kb::Ext>> chip_evals(n_chips);
    {
        Reader r{zc_blob.data(), zlen};
        const uint64_t np = r.u64();
        for (uint64_t q = 0; q < np; q++) { const uint64_t nc = r.u64(); r.skip(nc * 16); }
        r.skip(16);
        SP1HIP_REQUIRE(r.u64() == (uint64_t)L, "internal: zerocheck point dimension");
        for (auto& z : z_row) z = r.ext();
        r.skip(16);
        SP1HIP_REQUIRE(r.o == zc_sumcheck_size, "internal: zerocheck sumcheck size");
        SP1HIP_REQUIRE(r.u64() == (uint64_t)n_chips, "internal: zerocheck chip count");
        for (int c = 0; c < n_chips; c++) {
            const uint64_t w = r.u64();
            SP1HIP_REQUIRE(w == (uint64_t)chips[c].main_width + chips[c].prep_width, "internal: zerocheck opening width");
            for (uint64_t k = 0; k < w; k++) chip_evals[c].push_back(r.ext());
            prep_claims.insert(prep_claims.end(), chip_evals[c].begin(), chip_evals[c].begin() + chips[c].prep_width);
            main_claims.insert(main_claims.end(), chip_evals[c].begin() + chips[c].prep_width, chip_evals[c].end());
        }
    }
    sh_t[3] = std::chrono::steady_clock::now();
    marks.next("evaluation_proof");
    // ---- jagged evaluation proof over [preprocessed round, main round]
    std::vector<kb::Ext> claims = prep_claims;
    claims.insert(claims.end(), main_claims.begin(), main_claims.end());
    const size_t per_round[2] = {prep_claims.size(), main_claims.size()};
    sp1hip_stacked_data_t* rounds[2] = {preprocessed, main_data};
    // the evaluation proof is the last field of ShardProof and 90 % of its bytes: written straight into the caller's buffer,
    // behind the head that is assembled below (every size is known from the shapes)
    SP1HIP_REQUIRE(need >= jag_size, "internal error: shard proof sizes");
    uint8_t* const jag_at = h_proof + (need - jag_size);
    size_t jlen = jag_size;
    SP1HIP_TRY(sp1hip_jagged_prove(reinterpret_cast<const sp1hip_ext_t*>(z_row.data()), L, rounds, 2,
                                   reinterpret_cast<const sp1hip_ext_t*>(claims.data()), per_round, params.fri, ch, jag_at, &jlen,
                                   stream));
    SP1HIP_REQUIRE(jlen == jag_size, "internal error: jagged proof size");
    (void)s;

    // ---- bincode(ShardProof)
    sh_t[4] = std::chrono::steady_clock::now();
    marks.next("proof_bytes");
    std::vector<uint8_t> out;
    out.reserve(need - jag_size);
    put_u64(out, n_publics);
    for (int i = 0; i < n_publics; i++) put_felt(out, h_publics[i]);
    for (int k = 0; k < 8; k++) put_felt(out, main_commit[k]);
    out.insert(out.end(), gkr_blob.begin(), gkr_blob.begin() + glen);
    out.insert(out.end(), zc_blob.begin(), zc_blob.begin() + zc_sumcheck_size);
    put_u64(out, n_chips);
    for (int c = 0; c < n_chips; c++) {
        const size_t nl = strlen(chips[c].name);
        put_u64(out, nl);
        out.insert(out.end(), chips[c].name, chips[c].name + nl);
        put_u64(out, chips[c].prep_width);
        for (uint32_t k = 0; k < chips[c].prep_width; k++) for (int q = 0; q < 4; q++) put_felt(out, chip_evals[c][k].c[q]);
        put_u64(out, chips[c].main_width);
        for (uint32_t k = 0; k < chips[c].main_width; k++) for (int q = 0; q < 4; q++) put_felt(out, chip_evals[c][chips[c].prep_width + k].c[q]);
        put_u64(out, L + 1);
        for (int b = L; b >= 0; b--) put_felt(out, ((chips[c].real_rows >> b) & 1) ? kb::R1 : 0u);
    }
    if (out.size() + jlen != need) {
        set_error("internal error: shard proof size %zu != expected %zu", out.size() + jlen, need);
        return SP1HIP_ERROR_RUNTIME;
    }
    memcpy(h_proof, out.data(), out.size());
    *proof_len = need;
    challenger_restore(challenger, ch);
    if (sh_timing) {
        sh_t[5] = std::chrono::steady_clock::now();
        const auto ms = [&](int a, int b) { return std::chrono::duration<double, std::milli>(sh_t[b] - sh_t[a]).count(); };
        uint64_t miss_n1 = 0, miss_b1 = 0;
        arena_miss_stats(&miss_n1, &miss_b1);
        fprintf(stderr, "[sp1hip shard] prologue %.3f ms | commit %.3f | LogUp-GKR %.3f | zerocheck %.3f | evaluation proof %.3f | proof bytes %.3f | %llu arena misses (%.1f MB)\n",
                std::chrono::duration<double, std::milli>(sh_t[0] - sh_entry).count(), ms(0, 1), ms(1, 2), ms(2, 3), ms(3, 4), ms(4, 5),
                (unsigned long long)(miss_n1 - miss_n0), (double)(miss_b1 - miss_b0) / 1e6);
    }
    wait_plan.ok = true;
    return SP1HIP_SUCCESS;
}


// ------------------------------------------------------------------------------------------- the AirProver slot
// `setup_from_preprocessed_data_and_traces` (shard.rs:L406-L429): commit the preprocessed traces, keep the commitment
// round (the proving key's PreprocessedData) next to the verifying key it defines.
struct sp1hip_pk_s {
    sp1hip_stacked_data_t* preprocessed = nullptr;
    sp1hip_vk_t vk{};
    sp1hip_shard_params_t params{};
    ~sp1hip_pk_s() { if (preprocessed) sp1hip_stacked_data_free(preprocessed); }
};

int sp1hip_setup(const sp1hip_table_t* preprocessed_tables, int n_tables, const uint32_t pc_start[3],
                 const uint32_t initial_global_cumulative_sum[14], uint32_t enable_untrusted_programs,
                 sp1hip_shard_params_t params, sp1hip_pk_t** out, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(preprocessed_tables && n_tables > 0 && pc_start && initial_global_cumulative_sum && out, "null argument");
    for (int i = 0; i < 3; i++) SP1HIP_REQUIRE(pc_start[i] < kb::P, "pc_start not reduced");
    for (int i = 0; i < 14; i++) SP1HIP_REQUIRE(initial_global_cumulative_sum[i] < kb::P, "cumulative sum not reduced");
    SP1HIP_REQUIRE(enable_untrusted_programs < kb::P, "flag not reduced");
    std::unique_ptr<sp1hip_pk_s> pk(new sp1hip_pk_s());
    SP1HIP_TRY(sp1hip_jagged_commit(preprocessed_tables, n_tables, params.max_log_row_count, params.log_stacking_height,
                                    params.batch_size, params.fri.log_blowup, pk->vk.preprocessed_commit, &pk->preprocessed, stream));
    memcpy(pk->vk.pc_start, pc_start, 12);
    memcpy(pk->vk.initial_global_cumulative_sum, initial_global_cumulative_sum, 56);
    pk->vk.enable_untrusted_programs = enable_untrusted_programs;
    pk->params = params;
    *out = pk.release();
    return SP1HIP_SUCCESS;
}

void sp1hip_pk_free(sp1hip_pk_t* pk) { delete pk; }

int sp1hip_pk_vk(const sp1hip_pk_t* pk, sp1hip_vk_t* out) {
    SP1HIP_REQUIRE(pk && out, "null argument");
    *out = pk->vk;
    return SP1HIP_SUCCESS;
}

// `MachineVerifyingKey::observe_into` (/root/reference/crates/hypercube/src/verifier/config.rs:L97-L112)
int sp1hip_vk_observe_into(const sp1hip_vk_t* vk, sp1hip_challenger_t* challenger) {
    SP1HIP_REQUIRE(vk && challenger, "null argument");
    for (int k = 0; k < 8; k++) challenger_observe(challenger, vk->preprocessed_commit[k]);
    for (int k = 0; k < 3; k++) challenger_observe(challenger, vk->pc_start[k]);
    for (int k = 0; k < 14; k++) challenger_observe(challenger, vk->initial_global_cumulative_sum[k]);     // x[7] then y[7]
    challenger_observe(challenger, vk->enable_untrusted_programs);
    for (int k = 0; k < 6; k++) challenger_observe(challenger, 0u);                                        // the padding
    return SP1HIP_SUCCESS;
}

// `AirProver::prove_shard_with_pk` (shard.rs:L321-L345) after trace generation: default challenger, vk.observe_into,
// prove_shard_with_data.
int sp1hip_prove_shard_with_pk(const sp1hip_pk_t* pk, const sp1hip_shard_chip_t* chips, int n_chips, const uint32_t* h_publics,
                               int n_publics, const uint32_t* pow_witnesses, int n_pow_witnesses, uint8_t* h_proof, size_t* proof_len,
                               sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(pk && proof_len, "null argument");
    sp1hip_challenger_t* ch = nullptr;
    SP1HIP_TRY(sp1hip_challenger_new(&ch));
    struct ChGuard { sp1hip_challenger_t* c; ~ChGuard() { sp1hip_challenger_free(c); } } guard{ch};
    SP1HIP_TRY(sp1hip_vk_observe_into(&pk->vk, ch));
    if (n_pow_witnesses) SP1HIP_TRY(sp1hip_challenger_inject_pow_witnesses(ch, pow_witnesses, n_pow_witnesses));
    return sp1hip_prove_shard(chips, n_chips, h_publics, n_publics, pk->preprocessed, pk->params, ch, h_proof, proof_len, stream);
}

}  // extern "C"
