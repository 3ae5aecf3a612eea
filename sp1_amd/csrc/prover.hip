// sp1_amd/csrc/prover.hip — host side of the backend above the kernels: the Fiat–Shamir transcript
// and the BaseFold prover, mirroring the reference's operator interface for this path
// (same names, argument meaning and order of transcript operations):
//
//   sp1hip::DuplexChallenger      `DuplexChallenger<KoalaBear, KoalaPerm, 16, 8>` via `IopCtx::Challenger`
//                                 (/root/reference/slop/crates/challenger/src/lib.rs:L25-L87; semantics as
//                                 restated in /root/reference/sp1-gpu/crates/sys/include/challenger/challenger.cuh:L13-L118)
//   sp1hip::BasefoldProverData    `BasefoldProverData` (/root/reference/slop/crates/basefold-prover/src/prover.rs:L25-L31)
//   sp1hip::commit_mles           `BasefoldProver::commit_mles`                  (prover.rs:L78-L99)
//   sp1hip::prove_trusted_mle_evaluations  `BasefoldProver::prove_trusted_mle_evaluations` (prover.rs:L102-L243)
//       -> FriCpuProver::batch / commit_phase_round (/root/reference/slop/crates/basefold-prover/src/fri.rs:L31-L129)
//   bincode writer                `BasefoldProof` (/root/reference/slop/crates/basefold/src/verifier.rs:L94-L116),
//                                 `MerkleTreeOpeningAndProof`/`MerkleTreeTcsProof` (/root/reference/slop/crates/merkle-tree/src/tcs.rs:L49-L91)
//
// The host keeps only the transcript (a few hundred field ops per round); every O(n) step is a
// kernel on the caller's stream. One host<->device sync per fold round (16 B + 32 B read back).
// PoW witnesses: the SMALLEST valid witness is returned (the reference's rayon `find_any` returns
// any valid one).
#include <atomic>
#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "device_ctx.hpp"
#include "round_sync.hpp"
#include "tensor_table.hpp"

namespace sp1hip {

int merkle_finish_tree(uint32_t*, int, uint32_t, uint32_t*, const DeviceCtx*, hipStream_t, const uint32_t* = nullptr, uint32_t* = nullptr, uint32_t = 0);
void leaf_hash_plan(const sp1hip_tensor_t* tensors, int n_tensors, std::vector<LeafPart>* parts);
int leaf_hash_part(const uint32_t* const* d_cols, uint32_t width, int k, int n_parts, uint32_t height, uint32_t* d_carry,
                   uint32_t* d_tree, const DeviceCtx* ctx, hipStream_t s);
int commit_ext_pairs(const uint32_t* d_cw, int lg_n, uint32_t* d_tree, uint32_t* d_root_and_commit, hipStream_t s,
                     const uint32_t* d_publish_extra = nullptr, uint32_t* h_publish_slot = nullptr, uint32_t publish_seq = 0);
int fold_round_async(const uint32_t* d_cw, int lg_c, const uint32_t* d_mle, int lg_m, const kb::Ext& beta, uint32_t* d_cw_out,
                     uint32_t* d_mle_out, const uint32_t* d_eq_next, uint32_t* d_zero_val, uint32_t* d_partial, hipStream_t s);
int open_ext_pairs(const uint32_t* d_cw, int lg_n, const uint32_t* d_indices, size_t n_idx, uint32_t* d_values, hipStream_t s);
int shift_indices(uint32_t* d_idx, size_t n, hipStream_t s);
struct FoldOpenDesc { const uint32_t* cw; const uint32_t* tree; uint32_t lg_c, vals_off, paths_off, pad; };   // basefold.hip
int open_fold_rounds(const FoldOpenDesc* d_descs, int n_rounds, int max_lg_c, const uint32_t* d_indices, size_t n_idx,
                     uint32_t* d_out, hipStream_t s);
int ext_fixed_at_zero_async(const uint32_t* d_mle, int lg_n, const uint32_t* d_eq, uint32_t* d_out, hipStream_t s);
int eq_prefix_tables_soa_async(const kb::Ext* h_point, int d, uint32_t* d_out, hipStream_t s);


// ---------------------------------------------------------------- transcript
struct DuplexChallenger {
    uint32_t state[16] = {0};
    uint32_t in[8];
    int n_in = 0;
    uint32_t out[8];
    int n_out = 0;
    uint32_t injected[4];            // proof-of-work witnesses to use instead of searching (canonical), see grind()
    int n_injected = 0, next_injected = 0;

    void duplexing() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        p2_host_permute(state);
        for (int i = 0; i < 8; i++) out[i] = state[i];
        n_out = 8;
    }
    void observe(uint32_t x) {
        n_out = 0;
        in[n_in++] = x;
        if (n_in == 8) duplexing();
    }
    void observe_slice(const uint32_t* x, size_t n) { for (size_t i = 0; i < n; i++) observe(x[i]); }
    void observe_ext(const kb::Ext& e) { for (int i = 0; i < 4; i++) observe(e.c[i]); }
    uint32_t sample() {
        if (n_in != 0 || n_out == 0) duplexing();
        return out[--n_out];
    }
    kb::Ext sample_ext() { kb::Ext e; for (int i = 0; i < 4; i++) e.c[i] = sample(); return e; }
    uint32_t sample_bits(int bits) { return kb::from_monty(sample()) & ((1u << bits) - 1); }
    bool check_witness(int bits, uint32_t w) { observe(w); return sample_bits(bits) == 0; }
};

// Each lane tests one candidate witness: overwrite slot `pos` of the pre-loaded sponge state, permute,
// look at state[7] (the first word `sample` pops). atomicMin keeps the smallest hit.
struct GrindBase { uint32_t w[16]; };       // the sponge state the candidates are written into, passed by value
__global__ __launch_bounds__(256) void grind_kernel(GrindBase base_state, int pos, uint32_t mask,
                                                    uint32_t first, uint32_t count,
                                                    const p2::RoundConstants* __restrict__ rc, uint32_t* result) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint32_t w = first + t;             // canonical candidate
    if (w >= kb::P) return;
    const uint32_t wm = kb::to_monty(w);
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = base_state.w[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = (i == pos) ? wm : s[i];
    p2::permute(s, *rc);
    if ((kb::from_monty(s[7]) & mask) == 0) atomicMin(result, w);
}

static int grind(DuplexChallenger& ch, int bits, uint32_t* witness_monty, hipStream_t s) {
    SP1HIP_REQUIRE(bits >= 0 && bits < 31, "bits out of range");
    // state as `check_witness` will see it: pending inputs overwrite state[0..n_in), witness at n_in
    uint32_t base[16];
    memcpy(base, ch.state, sizeof base);
    for (int i = 0; i < ch.n_in; i++) base[i] = ch.in[i];
    const int pos = ch.n_in;
    const uint32_t mask = (1u << bits) - 1;
    uint32_t found = 0xffffffffu;
    if (ch.next_injected < ch.n_injected) {       // the caller's witness (sp1hip_challenger_inject_pow_witnesses)
        found = ch.injected[ch.next_injected++];
        *witness_monty = kb::to_monty(found);
        if (!ch.check_witness(bits, *witness_monty)) {
            set_error("grind: the injected %d-bit proof-of-work witness %u is not valid at this point of the transcript", bits, found);
            return SP1HIP_ERROR_INVALID_ARGUMENT;
        }
        return SP1HIP_SUCCESS;
    }
    if (bits <= 8) {
        // expected <= 256 candidates: cheaper on the host than a launch + sync
        for (uint32_t w = 0; w < kb::P; w++) {
            uint32_t st[16];
            memcpy(st, base, sizeof st);
            st[pos] = kb::to_monty(w);
            p2_host_permute(st);
            if ((kb::from_monty(st[7]) & mask) == 0) { found = w; break; }
        }
    } else {
        const DeviceCtx* ctx;
        SP1HIP_TRY(get_device_ctx(&ctx));
        AsyncScratch buf;
        SP1HIP_TRY(buf.alloc(4, s));
        uint32_t* d_res = (uint32_t*)buf.p;
        GrindBase gb;
        memcpy(gb.w, base, sizeof base);
        Mailbox mb;
        SP1HIP_TRY(mb.init(s));
        SP1HIP_HIP(hipMemsetAsync(d_res, 0xff, 4, s));
        uint32_t batch = 1u << std::min(22, bits + 3);
        for (uint64_t first = 0; first < kb::P && found == 0xffffffffu; first += batch) {
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(batch, kb::P - first);
            hipLaunchKernelGGL(grind_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, gb, pos, mask, (uint32_t)first,
                               cnt, ctx->d_rc, d_res);
            SP1HIP_LAUNCH_CHECK();
            SP1HIP_TRY(mb.fetch(d_res, 1, &found));
        }
    }
    if (found == 0xffffffffu) { set_error("grind: no witness found"); return SP1HIP_ERROR_RUNTIME; }
    *witness_monty = kb::to_monty(found);
    bool ok = ch.check_witness(bits, *witness_monty);
    if (!ok) { set_error("grind: internal error, witness rejected by the host transcript"); return SP1HIP_ERROR_RUNTIME; }
    return SP1HIP_SUCCESS;
}

// ---------------------------------------------------------------- prover data
// Buffers come from the stream-keyed arena (runtime.hip): steady-state proving re-uses the same HBM
// blocks without touching the driver.
struct DeviceBuf {
    void* p = nullptr;
    hipStream_t s = nullptr;
    size_t n = 0;
    int alloc(size_t bytes, hipStream_t stream) {
        s = stream;
        n = bytes;
        return arena_alloc(&p, bytes, stream);
    }
    ~DeviceBuf() { arena_free(p, n, s); }
    DeviceBuf() = default;
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
    uint32_t* u32() const { return (uint32_t*)p; }
};

}  // namespace sp1hip

struct sp1hip_challenger_s { sp1hip::DuplexChallenger ch; };

struct sp1hip_basefold_data_s {
    int lg_n = 0, lg_blowup = 0;
    std::vector<sp1hip_tensor_t> mles;                    // caller-owned inputs [2^lg_n x w], column-major
    std::vector<std::unique_ptr<sp1hip::DeviceBuf>> cws;  // codewords [2^(lg_n+lg_blowup) x w]
    std::vector<sp1hip_tensor_t> cw_tensors;
    sp1hip::DeviceBuf tree;
    uint32_t root[8], commit[8];
    uint32_t total_width = 0;
    // The blocks go back to the free list of the stream that created them, which orders their reuse behind that stream's
    // work only. A handle that was also read on ANOTHER stream (a proving key's preprocessed commitment is opened by every
    // prover, each on its own stream) waits for the device before it lets go.
    std::atomic<bool> foreign_use{false};
    ~sp1hip_basefold_data_s() { if (foreign_use) (void)hipDeviceSynchronize(); }
};

namespace sp1hip {

// transcript hooks for the other translation units (zerocheck.hip)
void challenger_observe(sp1hip_challenger_t* ch, uint32_t x) { ch->ch.observe(x); }
kb::Ext challenger_sample_ext(sp1hip_challenger_t* ch) { return ch->ch.sample_ext(); }
void challenger_restore(sp1hip_challenger_t* dst, const sp1hip_challenger_t* src) { dst->ch = src->ch; }
// the sponge as 34 words (state[16], in[8], n_in, out[8], n_out): the image a device-resident transcript starts from /
// hands back (gkr.hip: a layer's sumcheck rounds chain on the GPU)
void challenger_export(const sp1hip_challenger_t* ch, uint32_t* w) {
    memcpy(w, ch->ch.state, 64);
    memcpy(w + 16, ch->ch.in, 32);
    w[24] = (uint32_t)ch->ch.n_in;
    memcpy(w + 25, ch->ch.out, 32);
    w[33] = (uint32_t)ch->ch.n_out;
}
void challenger_import(sp1hip_challenger_t* ch, const uint32_t* w) {
    memcpy(ch->ch.state, w, 64);
    memcpy(ch->ch.in, w + 16, 32);
    ch->ch.n_in = (int)w[24];
    memcpy(ch->ch.out, w + 25, 32);
    ch->ch.n_out = (int)w[33];
}

// bincode writer straight into the caller's proof buffer (the BaseFold proof of a core shard is 1.4 MB: building it in a
// vector and copying it out — here, then again in the jagged and the shard wrappers — was five copies of it per proof)
struct ByteWriter {
    uint8_t* p = nullptr;
    size_t cap = 0, n = 0;
    bool overflow = false;
    uint8_t* take(size_t k) {
        if (n + k > cap) { overflow = true; return nullptr; }
        uint8_t* q = p + n;
        n += k;
        return q;
    }
    void u64(uint64_t v) { if (uint8_t* q = take(8)) memcpy(q, &v, 8); }      // little-endian host
    void u32(uint32_t v) { if (uint8_t* q = take(4)) memcpy(q, &v, 4); }
    void felt(uint32_t monty) { u32(kb::from_monty(monty)); }
    void felts(const uint32_t* m, size_t k) {
        uint8_t* q = take(4 * k);
        if (!q) return;
        for (size_t i = 0; i < k; i++) {
            const uint32_t c = kb::from_monty(m[i]);     // canonical word == its four bytes
            memcpy(q + 4 * i, &c, 4);
        }
    }
    void ext(const kb::Ext& e) { felts(e.c, 4); }
    void canonical_words(const uint32_t* c, size_t k) {  // words the device has already taken out of Montgomery form
        if (uint8_t* q = take(4 * k)) memcpy(q, c, 4 * k);
    }
};

// values / paths: CANONICAL words (the query phase converts the whole opening buffer on the device before it leaves)
static void write_opening(ByteWriter& w, const uint32_t* values, size_t n_values, size_t n_idx, size_t width,
                          const uint32_t* root, size_t lg_h, const uint32_t* paths, size_t n_paths) {
    w.u64(n_values);
    w.canonical_words(values, n_values);
    w.u64(2); w.u64(n_idx); w.u64(width);
    w.felts(root, 8);
    w.u64(lg_h);
    w.u64(width);
    w.u64(n_idx * lg_h);
    w.canonical_words(paths, n_paths);
    w.u64(2); w.u64(n_idx); w.u64(lg_h);
}

static int log2_ceil(size_t x) { int l = 0; while (((size_t)1 << l) < x) l++; return l; }

static std::vector<kb::Ext> partial_lagrange_host(const std::vector<kb::Ext>& pt) {
    std::vector<kb::Ext> ev{kb::ext_one()};
    for (const kb::Ext& x : pt) {
        std::vector<kb::Ext> nx(ev.size() * 2);
        for (size_t i = 0; i < ev.size(); i++) {
            kb::Ext prod = kb::ext_mul(ev[i], x);
            nx[2 * i] = kb::ext_sub(ev[i], prod);
            nx[2 * i + 1] = prod;
        }
        ev.swap(nx);
    }
    return ev;
}

static size_t opening_size(size_t n_idx, size_t width, size_t lg_h) {
    return 8 + 4 * n_idx * width + 24 + 32 + 8 + 8 + 8 + 32 * n_idx * lg_h + 24;
}

static size_t proof_size(int dim, const std::vector<uint32_t>& widths, const sp1hip_fri_config_t& cfg) {
    size_t q = (size_t)cfg.num_queries;
    size_t sz = 8 + (size_t)dim * 32 + 8 + (size_t)dim * 32 + 8;
    for (uint32_t w : widths) sz += opening_size(q, w, (size_t)dim + cfg.log_blowup);
    sz += 8;
    for (int r = 0; r < dim; r++) sz += opening_size(q, 8, (size_t)dim + cfg.log_blowup - 1 - r);
    return sz + 16 + 4 + 4;
}

static int prove_trusted_mle_evaluations(std::vector<kb::Ext> point, sp1hip_basefold_data_s* const* rounds, int n_rounds,
                                         const kb::Ext* claims, size_t n_claims, const sp1hip_fri_config_t& cfg,
                                         DuplexChallenger& ch, uint8_t* out, size_t out_cap, size_t* out_len, hipStream_t s) {
    const int dim = (int)point.size();
    const int lb = cfg.log_blowup;
    const size_t nq = (size_t)cfg.num_queries;
    // all mles of all rounds, in order
    std::vector<sp1hip_tensor_t> mles;
    for (int r = 0; r < n_rounds; r++) {
        SP1HIP_REQUIRE(rounds[r]->lg_n == dim, "eval point dimension mismatch");
        SP1HIP_REQUIRE(rounds[r]->lg_blowup == lb, "round committed with a different blowup");
        for (auto& m : rounds[r]->mles) mles.push_back(m);
    }
    size_t total_len = 0;
    for (auto& m : mles) total_len += m.width;
    SP1HIP_REQUIRE(total_len == n_claims, "one evaluation claim per committed column expected");
    SP1HIP_REQUIRE(dim >= 1, "at least one variable expected");
    SP1HIP_REQUIRE(dim + lb <= kb::TWO_ADICITY, "instance exceeds two-adicity");

    ByteWriter w;
    w.p = out;
    w.cap = out_cap;
    // Grind for batch randomness, then the batching coefficients.
    uint32_t batch_witness;
    SP1HIP_TRY(grind(ch, 5, &batch_witness, s));
    std::vector<kb::Ext> bpt(log2_ceil(total_len));
    for (auto& x : bpt) x = ch.sample_ext();
    std::vector<kb::Ext> coeffs = partial_lagrange_host(bpt);

    const size_t n = (size_t)1 << dim, N0 = n << lb;
    DeviceBuf d_coeffs, d_mle[2], d_eq;
    SP1HIP_TRY(d_coeffs.alloc(total_len * 16, s));
    SP1HIP_TRY(d_mle[0].alloc(n * 16, s));
    SP1HIP_TRY(d_mle[1].alloc(n * 8 + 16, s));
    SP1HIP_TRY(d_eq.alloc(n * 16, s));      // every prefix table of eq(point[0..t), .), t < dim: round r reads table dim - r - 1
    Mailbox mb;                                   // every device -> host hand-over below goes through it (round_sync.hpp)
    SP1HIP_TRY(mb.init(s));
    PinnedStage stage;
    SP1HIP_TRY(stage.init(s));
    SP1HIP_TRY(stage.upload(d_coeffs.p, coeffs.data(), total_len * 16));
    SP1HIP_TRY(sp1hip_basefold_batch(mles.data(), (int)mles.size(), dim, d_coeffs.u32(), d_mle[0].u32(), s));
    kb::Ext cur_claim = kb::ext_zero();
    for (size_t i = 0; i < n_claims; i++) cur_claim = kb::ext_add(cur_claim, kb::ext_mul(claims[i], coeffs[i]));

    // codewords of every round are kept for the query phase: sizes N0, N0/2, ..., 2 (ext SoA)
    std::vector<std::unique_ptr<DeviceBuf>> cws, trees;
    cws.emplace_back(new DeviceBuf());
    SP1HIP_TRY(cws.back()->alloc(N0 * 16, s));
    SP1HIP_TRY(sp1hip_rs_encode_batch(cws.back()->u32(), d_mle[0].u32(), dim, lb, 4, s));

    ch.observe(kb::to_monty((uint32_t)dim));
    DeviceBuf d_rb;  // [0..4) zero_val, [4..20) root+commit, [20..24) final poly
    SP1HIP_TRY(d_rb.alloc(24 * 4, s));
    std::vector<std::array<uint32_t, 8>> round_roots;
    std::vector<kb::Ext> uni;
    std::vector<std::array<uint32_t, 8>> fri_commitments;
    int cur = 0;
    SP1HIP_TRY(eq_prefix_tables_soa_async(point.data(), dim - 1, d_eq.u32(), s));
    auto eq_table = [&](int t) { return d_eq.u32() + 4 * (((size_t)1 << t) - 1); };     // eq over the first t coordinates
    // every round's codeword, tree and the fold scratch are allocated here: between a round's hand-over and its launches
    // the host does nothing but enqueue (the GPU was waiting 10-16 us per launch behind the allocations)
    DeviceBuf d_fold_partial;
    SP1HIP_TRY(d_fold_partial.alloc((((size_t)n / 2 + 255) / 256) * 16, s));
    for (int r = 0; r < dim; r++) {
        const int lg_c = dim - r + lb;
        trees.emplace_back(new DeviceBuf());
        SP1HIP_TRY(trees.back()->alloc((((size_t)2 << (lg_c - 1)) - 1) * 32, s));
        cws.emplace_back(new DeviceBuf());
        SP1HIP_TRY(cws.back()->alloc(((size_t)1 << (lg_c - 1)) * 16, s));
    }
    // zero_val of round 0 = sum_i eq(point', i) * mle[2 i]; every later round's comes out of the fold before it
    SP1HIP_TRY(ext_fixed_at_zero_async(d_mle[cur].u32(), dim, eq_table(dim - 1), d_rb.u32(), s));
    for (int r = 0; r < dim; r++) {
        const int lg_m = dim - r;             // current mle has 2^lg_m entries
        const int lg_c = lg_m + lb;           // current codeword has 2^lg_c entries
        kb::Ext last = point.back();
        point.pop_back();
        // commit to the paired leaves of the current codeword; the tree's last kernel hands [zero_val | root | commitment]
        // to the host
        SP1HIP_TRY(commit_ext_pairs(cws[r]->u32(), lg_c, trees[r]->u32(), d_rb.u32() + 4, s, d_rb.u32(), mb.h_slot, mb.seq + 1));
        uint32_t rb[20];
        SP1HIP_TRY(mb.wait_next(rb, 20));
        kb::Ext zero_val{{rb[0], rb[1], rb[2], rb[3]}};
        kb::Ext one_val = kb::ext_add(kb::ext_mul(kb::ext_sub(cur_claim, zero_val), kb::ext_inv(last)), zero_val);
        uni.push_back(zero_val);
        uni.push_back(one_val);
        ch.observe_ext(zero_val);
        ch.observe_ext(one_val);
        std::array<uint32_t, 8> root, commit;
        memcpy(root.data(), rb + 4, 32);
        memcpy(commit.data(), rb + 12, 32);
        round_roots.push_back(root);
        fri_commitments.push_back(commit);
        ch.observe_slice(commit.data(), 8);
        kb::Ext beta = ch.sample_ext();
        // both folds and the next round's zero_val partials in one launch
        SP1HIP_TRY(fold_round_async(cws[r]->u32(), lg_c, d_mle[cur].u32(), lg_m, beta, cws[r + 1]->u32(),
                                    d_mle[cur ^ 1].u32(), lg_m >= 2 ? eq_table(lg_m - 2) : nullptr, d_rb.u32(), d_fold_partial.u32(), s));
        cur ^= 1;
        cur_claim = kb::ext_add(zero_val, kb::ext_mul(beta, one_val));
    }
    // final_poly = first ext element of the last codeword (length 2^lb)
    {
        const size_t len = (size_t)1 << lb;
        for (int k = 0; k < 4; k++)
            SP1HIP_HIP(hipMemcpyAsync(d_rb.u32() + 20 + k, cws.back()->u32() + (size_t)k * len, 4, hipMemcpyDeviceToDevice, s));
    }
    uint32_t fp[4];
    SP1HIP_TRY(mb.fetch(d_rb.u32() + 20, 4, fp));
    kb::Ext final_poly{{fp[0], fp[1], fp[2], fp[3]}};
    ch.observe_ext(final_poly);
    uint32_t pow_witness;
    SP1HIP_TRY(grind(ch, cfg.proof_of_work_bits, &pow_witness, s));
    std::vector<uint32_t> q(nq);
    for (auto& x : q) x = ch.sample_bits(dim + lb);

    // ---- serialise: univariate messages, commitments, openings
    w.u64((uint64_t)dim);
    for (auto& e : uni) w.ext(e);
    w.u64((uint64_t)dim);
    for (auto& c : fri_commitments) w.felts(c.data(), 8);

    // Query phase: every opening (component rounds, then the dim fold rounds) is produced into ONE device buffer and
    // brought back with one copy and one synchronise instead of one round trip per opening.
    DeviceBuf d_idx, d_open;
    struct Slot { size_t vals_off, n_vals, paths_off, n_paths; };
    std::vector<Slot> slots;
    size_t words = 0;
    for (int r = 0; r < n_rounds; r++) {
        const size_t nv = nq * rounds[r]->total_width, np = nq * (size_t)(dim + lb) * 8;
        slots.push_back(Slot{words, nv, words + nv, np});
        words += nv + np;
    }
    for (int r = 0; r < dim; r++) {
        const size_t lg_h = (size_t)(dim + lb - r - 1);
        const size_t nv = nq * 8, np = nq * lg_h * 8;
        slots.push_back(Slot{words, nv, words + nv, np});
        words += nv + np;
    }
    SP1HIP_TRY(d_idx.alloc(nq * 4, s));
    SP1HIP_TRY(d_open.alloc(std::max<size_t>(words, 1) * 4, s));
    SP1HIP_TRY(stage.upload(d_idx.p, q.data(), nq * 4));
    for (int r = 0; r < n_rounds; r++) {
        sp1hip_basefold_data_s* pd = rounds[r];
        const Slot& sl = slots[r];
        SP1HIP_TRY(sp1hip_merkle_open(pd->cw_tensors.data(), (int)pd->cw_tensors.size(), dim + lb, pd->tree.u32(), d_idx.u32(),
                                      nq, d_open.u32() + sl.vals_off, d_open.u32() + sl.paths_off, s));
    }
    {   // every fold round's pairs and paths in one launch
        SP1HIP_REQUIRE(words < ((size_t)1 << 32), "opening buffer too large");
        std::vector<FoldOpenDesc> descs(dim);
        for (int r = 0; r < dim; r++) {
            const Slot& sl = slots[n_rounds + r];
            descs[r] = FoldOpenDesc{cws[r]->u32(), trees[r]->u32(), (uint32_t)(dim + lb - r), (uint32_t)sl.vals_off, (uint32_t)sl.paths_off, 0u};
        }
        DeviceBuf d_descs;
        SP1HIP_TRY(d_descs.alloc(descs.size() * sizeof(FoldOpenDesc), s));
        SP1HIP_TRY(stage.upload(d_descs.p, descs.data(), descs.size() * sizeof(FoldOpenDesc)));
        SP1HIP_TRY(open_fold_rounds(reinterpret_cast<const FoldOpenDesc*>(d_descs.p), dim, dim + lb, d_idx.u32(), nq, d_open.u32(), s));
    }
    // The opening buffer leaves the device as CANONICAL words (one conversion launch instead of ~340k host reductions) and
    // lands in a pinned block when it fits one (no pageable bounce, no stream synchronise: the mailbox fence below orders
    // the host behind the copy).
    SP1HIP_TRY(sp1hip_from_monty(d_open.u32(), words, (sp1hip_stream_t)s));
    std::vector<uint32_t> opened_pageable;
    PinnedBlock dl{nullptr};
    struct Release { PinnedBlock* b; ~Release() { if (b->h) pinned_stage_release(*b); } } release{&dl};
    const uint32_t* opened = nullptr;
    if (words * 4 <= PINNED_STAGE_BYTES && pinned_stage_acquire(&dl) == SP1HIP_SUCCESS) {
        SP1HIP_HIP(hipMemcpyAsync(dl.h, d_open.p, words * 4, hipMemcpyDeviceToHost, s));
        SP1HIP_TRY(mb.fetch(nullptr, 0, nullptr));             // (its completion also covers the q upload above)
        opened = reinterpret_cast<const uint32_t*>(dl.h);
    } else {
        dl.h = nullptr;
        opened_pageable.resize(std::max<size_t>(words, 1));
        SP1HIP_HIP(hipMemcpyAsync(opened_pageable.data(), d_open.p, words * 4, hipMemcpyDeviceToHost, s));
        SP1HIP_HIP(hipStreamSynchronize(s));
        opened = opened_pageable.data();
    }
    w.u64((uint64_t)n_rounds);
    for (int r = 0; r < n_rounds; r++) {
        const Slot& sl = slots[r];
        write_opening(w, opened + sl.vals_off, sl.n_vals, nq, rounds[r]->total_width, rounds[r]->root, (size_t)(dim + lb), opened + sl.paths_off, sl.n_paths);
    }
    w.u64((uint64_t)dim);
    for (int r = 0; r < dim; r++) {
        const Slot& sl = slots[n_rounds + r];
        write_opening(w, opened + sl.vals_off, sl.n_vals, nq, 8, round_roots[r].data(), (size_t)(dim + lb - r - 1), opened + sl.paths_off, sl.n_paths);
    }
    w.ext(final_poly);
    w.felt(pow_witness);
    w.felt(batch_witness);
    SP1HIP_REQUIRE(!w.overflow, "internal error: BaseFold proof larger than its computed size");
    *out_len = w.n;
    return SP1HIP_SUCCESS;
}

}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_challenger_new(sp1hip_challenger_t** out) {
    SP1HIP_REQUIRE(out, "null output");
    *out = new sp1hip_challenger_s();
    return SP1HIP_SUCCESS;
}
int sp1hip_challenger_clone(const sp1hip_challenger_t* ch, sp1hip_challenger_t** out) {
    SP1HIP_REQUIRE(ch && out, "null argument");
    *out = new sp1hip_challenger_s(*ch);
    return SP1HIP_SUCCESS;
}
void sp1hip_challenger_free(sp1hip_challenger_t* ch) { delete ch; }
int sp1hip_challenger_observe(sp1hip_challenger_t* ch, const uint32_t* felts, size_t n) {
    SP1HIP_REQUIRE(ch && (felts || n == 0), "null argument");
    for (size_t i = 0; i < n; i++) SP1HIP_REQUIRE(felts[i] < kb::P, "non-reduced field word");
    ch->ch.observe_slice(felts, n);
    return SP1HIP_SUCCESS;
}
int sp1hip_challenger_sample(sp1hip_challenger_t* ch, uint32_t* out) {
    SP1HIP_REQUIRE(ch && out, "null argument");
    *out = ch->ch.sample();
    return SP1HIP_SUCCESS;
}
int sp1hip_challenger_sample_ext(sp1hip_challenger_t* ch, sp1hip_ext_t* out) {
    SP1HIP_REQUIRE(ch && out, "null argument");
    kb::Ext e = ch->ch.sample_ext();
    memcpy(out->c, e.c, 16);
    return SP1HIP_SUCCESS;
}
int sp1hip_challenger_sample_bits(sp1hip_challenger_t* ch, int bits, uint32_t* out) {
    SP1HIP_REQUIRE(ch && out && bits >= 0 && bits < 32, "bad argument");
    *out = ch->ch.sample_bits(bits);
    return SP1HIP_SUCCESS;
}
int sp1hip_challenger_check_witness(sp1hip_challenger_t* ch, int bits, uint32_t witness, int* ok) {
    SP1HIP_REQUIRE(ch && ok && bits >= 0 && bits < 32 && witness < kb::P, "bad argument");
    *ok = ch->ch.check_witness(bits, witness) ? 1 : 0;
    return SP1HIP_SUCCESS;
}
int sp1hip_challenger_grind(sp1hip_challenger_t* ch, int bits, uint32_t* witness, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(ch && witness, "null argument");
    return grind(ch->ch, bits, witness, S(stream));
}
int sp1hip_challenger_inject_pow_witnesses(sp1hip_challenger_t* ch, const uint32_t* witnesses, int n) {
    SP1HIP_REQUIRE(ch && n >= 0 && n <= 4 && (witnesses || n == 0), "bad argument");
    for (int i = 0; i < n; i++) SP1HIP_REQUIRE(witnesses[i] < kb::P, "witness not a canonical field element");
    for (int i = 0; i < n; i++) ch->ch.injected[i] = witnesses[i];
    ch->ch.n_injected = n;
    ch->ch.next_injected = 0;
    return SP1HIP_SUCCESS;
}
int sp1hip_challenger_state(const sp1hip_challenger_t* ch, uint32_t* out34) {
    SP1HIP_REQUIRE(ch && out34, "null argument");
    memset(out34, 0, 34 * 4);
    memcpy(out34, ch->ch.state, 64);
    out34[16] = (uint32_t)ch->ch.n_in;
    memcpy(out34 + 17, ch->ch.in, 4 * ch->ch.n_in);
    out34[25] = (uint32_t)ch->ch.n_out;
    memcpy(out34 + 26, ch->ch.out, 4 * ch->ch.n_out);
    return SP1HIP_SUCCESS;
}

}  // extern "C"

namespace sp1hip {
// sp1hip_commit_mles with a hook: `before_encode(i, stream)` is called right before message i is encoded, on the stream
// that encodes it — the stacked commit fills message i's slice of its dense buffer there (stacked.hip), so that with the
// encodes on the side stream the 1.6 GB of table -> dense copies of a core shard run under the leaf hashes of the
// previous messages instead of in front of the whole commitment.
int commit_mles_hooked(const sp1hip_tensor_t* mles, int n_mles, int lg_n, int lg_blowup, uint32_t h_commit[8],
                       sp1hip_basefold_data_t** out, sp1hip_stream_t stream, const std::function<int(int, hipStream_t)>* before_encode) {
    SP1HIP_REQUIRE(mles && n_mles > 0 && out && h_commit, "bad argument");
    SP1HIP_REQUIRE(lg_n >= 0 && lg_blowup >= 0 && lg_n + lg_blowup <= kb::TWO_ADICITY, "size out of range");
    hipStream_t s = S(stream);
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));   // also configures the memory pool before the first allocation
    std::unique_ptr<sp1hip_basefold_data_s> pd(new sp1hip_basefold_data_s());
    pd->lg_n = lg_n;
    pd->lg_blowup = lg_blowup;
    const int lg_h = lg_n + lg_blowup;
    const size_t N = (size_t)1 << lg_h;
    for (int i = 0; i < n_mles; i++) {
        SP1HIP_REQUIRE(mles[i].d_data || mles[i].width == 0, "null mle");
        pd->mles.push_back(mles[i]);
        pd->cws.emplace_back(new DeviceBuf());
        SP1HIP_TRY(pd->cws.back()->alloc(N * mles[i].width * 4, s));
        pd->cw_tensors.push_back({pd->cws.back()->u32(), mles[i].width});
        pd->total_width += mles[i].width;
    }
    SP1HIP_TRY(pd->tree.alloc((2 * N - 1) * 32, s));
    DeviceBuf rc;
    SP1HIP_TRY(rc.alloc(64, s));
    const char* ov = getenv("SP1HIP_COMMIT_OVERLAP");
    // default: overlap when the codeword is large enough to fill the chip; "0" never, "1" always (tests)
    const bool overlap = ov ? ov[0] != '0' : N * (size_t)pd->total_width >= ((size_t)1 << 24);
    std::vector<LeafPart> parts;
    if (overlap) leaf_hash_plan(pd->cw_tensors.data(), n_mles, &parts);
    if (parts.size() >= 2) {
        // Leaf hashing is VALU-issue bound, the encode passes wait on HBM a quarter of their time: encode tensor
        // k + 1 on the side stream while this stream absorbs tensor k into the per-row sponge states.
        hipStream_t aux;
        hipEvent_t* ev;
        SP1HIP_TRY(aux_stream_for(s, n_mles + 1, &aux, &ev));
        struct Join {                      // an early return must not hand buffers back while `aux` still writes them
            hipStream_t aux;
            ~Join() { (void)hipStreamSynchronize(aux); }
        };
        TensorTable tab;
        uint32_t tw;
        SP1HIP_TRY(make_tensor_table(pd->cw_tensors.data(), n_mles, &tab, &tw));
        DeviceBuf cols, carry;
        SP1HIP_TRY(cols.alloc((size_t)tw * sizeof(uint32_t*), s));
        SP1HIP_TRY(carry.alloc(N * 8 * 4, s));
        SP1HIP_TRY(expand_columns_async(tab, tw, N, (const uint32_t**)cols.p, s));
        SP1HIP_HIP(hipEventRecord(ev[n_mles], s));            // inputs and recycled buffers are ordered on `s`
        SP1HIP_HIP(hipStreamWaitEvent(aux, ev[n_mles], 0));
        {
            Join join{aux};
            for (int i = 0; i < n_mles; i++) {
                if (before_encode) SP1HIP_TRY((*before_encode)(i, aux));
                SP1HIP_TRY(sp1hip_rs_encode_batch(pd->cws[i]->u32(), mles[i].d_data, lg_n, lg_blowup, mles[i].width, aux));
                SP1HIP_HIP(hipEventRecord(ev[i], aux));
            }
            for (int k = 0; k < (int)parts.size(); k++) {
                SP1HIP_HIP(hipStreamWaitEvent(s, ev[parts[k].last_tensor], 0));
                SP1HIP_TRY(leaf_hash_part((const uint32_t* const*)cols.p + parts[k].c0, parts[k].width, k, (int)parts.size(),
                                          (uint32_t)N, carry.u32(), pd->tree.u32(), ctx, s));
            }
            SP1HIP_TRY(merkle_finish_tree(pd->tree.u32(), lg_h, tw, rc.u32(), ctx, s));
            uint32_t h[16];
            Mailbox mb;
            SP1HIP_TRY(mb.init(s));
            SP1HIP_TRY(mb.fetch(rc.p, 16, h));              // `s` waited for every encode: `aux` is idle here
            memcpy(pd->root, h, 32);
            memcpy(pd->commit, h + 8, 32);
        }
        memcpy(h_commit, pd->commit, 32);
        *out = pd.release();
        return SP1HIP_SUCCESS;
    }
    for (int i = 0; i < n_mles; i++) {
        if (before_encode) SP1HIP_TRY((*before_encode)(i, s));
        SP1HIP_TRY(sp1hip_rs_encode_batch(pd->cws[i]->u32(), mles[i].d_data, lg_n, lg_blowup, mles[i].width, s));
    }
    SP1HIP_TRY(sp1hip_merkle_commit(pd->cw_tensors.data(), n_mles, lg_h, pd->tree.u32(), rc.u32(), s));
    uint32_t h[16];
    Mailbox mb;
    SP1HIP_TRY(mb.init(s));
    SP1HIP_TRY(mb.fetch(rc.p, 16, h));
    memcpy(pd->root, h, 32);
    memcpy(pd->commit, h + 8, 32);
    memcpy(h_commit, pd->commit, 32);
    *out = pd.release();
    return SP1HIP_SUCCESS;
}
}  // namespace sp1hip

extern "C" {

int sp1hip_commit_mles(const sp1hip_tensor_t* mles, int n_mles, int lg_n, int lg_blowup, uint32_t h_commit[8],
                       sp1hip_basefold_data_t** out, sp1hip_stream_t stream) {
    return commit_mles_hooked(mles, n_mles, lg_n, lg_blowup, h_commit, out, stream, nullptr);
}

void sp1hip_basefold_data_free(sp1hip_basefold_data_t* data) { delete data; }

int sp1hip_basefold_data_codeword(const sp1hip_basefold_data_t* data, int k, const uint32_t** d_cw, uint32_t* width,
                                  int* lg_height) {
    SP1HIP_REQUIRE(data && k >= 0 && k < (int)data->cws.size(), "bad argument");
    if (d_cw) *d_cw = data->cws[k]->u32();
    if (width) *width = data->mles[k].width;
    if (lg_height) *lg_height = data->lg_n + data->lg_blowup;
    return SP1HIP_SUCCESS;
}
int sp1hip_basefold_data_tree(const sp1hip_basefold_data_t* data, const uint32_t** d_tree, int* lg_height) {
    SP1HIP_REQUIRE(data, "null argument");
    if (d_tree) *d_tree = data->tree.u32();
    if (lg_height) *lg_height = data->lg_n + data->lg_blowup;
    return SP1HIP_SUCCESS;
}

size_t sp1hip_basefold_proof_size(int dim, const uint32_t* round_widths, int n_rounds, sp1hip_fri_config_t config) {
    std::vector<uint32_t> w(round_widths, round_widths + n_rounds);
    return proof_size(dim, w, config);
}

int sp1hip_basefold_prove(const sp1hip_ext_t* h_point, int dim, sp1hip_basefold_data_t* const* rounds, int n_rounds,
                          const sp1hip_ext_t* h_claims, size_t n_claims, sp1hip_fri_config_t config,
                          sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(h_point && rounds && n_rounds > 0 && h_claims && challenger && proof_len, "null argument");
    SP1HIP_REQUIRE(dim >= 1 && dim <= kb::TWO_ADICITY, "dim out of range");
    SP1HIP_REQUIRE(config.num_queries > 0 && config.log_blowup >= 0 && config.proof_of_work_bits >= 0, "bad config");
    std::vector<uint32_t> widths;
    for (int r = 0; r < n_rounds; r++) {
        SP1HIP_REQUIRE(rounds[r], "null round");
        widths.push_back(rounds[r]->total_width);
        if (rounds[r]->tree.s != S(stream)) rounds[r]->foreign_use = true;
    }
    const size_t need = proof_size(dim, widths, config);
    if (!h_proof || *proof_len < need) {
        *proof_len = need;
        set_error("sp1hip_basefold_prove: proof buffer too small, need %zu bytes", need);
        return SP1HIP_ERROR_BUFFER_TOO_SMALL;
    }
    std::vector<kb::Ext> point(dim);
    memcpy(point.data(), h_point, (size_t)dim * 16);
    DuplexChallenger ch = challenger->ch;  // commit to the transcript only on success
    size_t written = 0;
    SP1HIP_TRY(prove_trusted_mle_evaluations(point, rounds, n_rounds, reinterpret_cast<const kb::Ext*>(h_claims), n_claims,
                                             config, ch, h_proof, need, &written, S(stream)));      // written in place
    if (written != need) {
        set_error("internal error: proof size %zu != expected %zu", written, need);
        return SP1HIP_ERROR_RUNTIME;
    }
    *proof_len = written;
    challenger->ch = ch;
    return SP1HIP_SUCCESS;
}

}  // extern "C"
