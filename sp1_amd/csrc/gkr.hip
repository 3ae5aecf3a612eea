// sp1_amd/csrc/gkr.hip — LogUp-GKR on the device: the lookup argument that sits between
// `commit_traces` and zerocheck in a shard proof.
//
//   sp1hip_logup_gkr_prove  `GkrProverImpl::prove_logup_gkr`   /root/reference/crates/hypercube/src/logup_gkr/prover.rs:L70-L215
//     first layer           `generate_interaction_vals` / `generate_first_layer`   execution.rs:L13-L36, L112-L252
//     circuit               `layer_transition`, `extract_outputs`                 execution.rs:L38-L110, L254-L382
//     per-layer sumcheck    `prove_gkr_round` + `LogupRoundPolynomial`            cpu.rs:L146-L226, logup_poly.rs:L70-L553
//                           driven as `reduce_sumcheck_to_evaluation`             /root/reference/slop/crates/sumcheck/src/prover.rs:L13-L96
//   Output: bincode(LogupGkrProof) (/root/reference/crates/hypercube/src/logup_gkr/proof.rs:L32-L62).
//
// MI355X shape
//  * Layout: every (chip, interaction) owns contiguous per-row vectors `N[level][i][row]`, `D[level][i][row]`
//    over the chip's REAL rows only (padding rows are the constants (0, 1) and are never stored): a wave
//    walks consecutive rows of one interaction, so all loads are unit-stride (4 B lanes for the base-field
//    first-layer numerators, 16 B lanes for ext), and the interaction's program / eq weight are wave-uniform.
//  * The fraction tree (level L -> 1) is built once and kept: level l has ceil(h / 2^(L-l)) rows per chip.
//    The GKR layer with v row variables reads level v+1 in place (numerator_0/1 = even/odd rows).
//  * One fused kernel per sumcheck round over a row variable: fold the previous round's four tables with
//    alpha and accumulate the next round's three sums (y(0), 8 y(1/2), eq mass of the real entries) from
//    the folded values in registers. The padding rows enter in closed form on the host, exactly as the
//    reference does it (`eq_correction_term`, logup_poly.rs:L521-L530).
//  * eq over the row variables is never folded: the per-round tables are the partial-Lagrange tables of
//    the remaining prefix of the point, built for all prefix lengths by one launch per layer; the factor
//    of the already-bound variables is a host scalar.
//  * Once the row variables are bound, a layer is 2^niv x 4 values: the interaction-variable rounds run on
//    the host (a few hundred ext products).
#include <algorithm>
#include <array>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "device_ctx.hpp"
#include "host_par.hpp"
#include "round_sync.hpp"
#include "tensor_table.hpp"

namespace sp1hip {

void challenger_observe(sp1hip_challenger_t* ch, uint32_t x);
kb::Ext challenger_sample_ext(sp1hip_challenger_t* ch);
void challenger_restore(sp1hip_challenger_t* dst, const sp1hip_challenger_t* src);
void challenger_export(const sp1hip_challenger_t* ch, uint32_t* w34);
void challenger_import(sp1hip_challenger_t* ch, const uint32_t* w34);

namespace gkr {

using Ext = kb::Ext;
struct DeviceBuf : AsyncScratch {
    uint32_t* u32() const { return (uint32_t*)p; }
    Ext* ext() const { return (Ext*)p; }
};

__device__ __forceinline__ Ext ld_ext(const Ext* p, uint32_t i) {
    const q4_t v = gptr(reinterpret_cast<const q4_t*>(p))[i];
    return Ext{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void st_ext(Ext* p, uint32_t i, const Ext& e) {
    const q4_t v = {e.c[0], e.c[1], e.c[2], e.c[3]};
    gptr(reinterpret_cast<q4_t*>(p))[i] = v;
}

// ---- block reduction of NS ext accumulators -> partials[block][4 NS]
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, __shfl_down(v, off, 64));
    return v;
}
template <int NS>
__device__ __forceinline__ void block_reduce_store(const Ext (&acc)[NS], uint32_t* out) {
    __shared__ uint32_t sm[4][4 * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = wave_sum(acc[s].c[k]);
            if (lane == 0) sm[wave][4 * s + k] = w;
        }
    __syncthreads();
    if (threadIdx.x < 4 * NS) {
        uint32_t a = 0;
        for (int i = 0; i < 4; i++) a = kb::add(a, sm[i][threadIdx.x]);
        out[threadIdx.x] = a;
    }
}
template <int NS>
__global__ __launch_bounds__(256) void reduce_partials(const uint32_t* __restrict__ partials, uint32_t n, uint32_t* out) {
    Ext acc[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) acc[s] = kb::ext_zero();
    for (uint32_t i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t* q = partials + ((size_t)i * NS + s) * 4;
            acc[s] = kb::ext_add(acc[s], Ext{{q[0], q[1], q[2], q[3]}});
        }
    block_reduce_store<NS>(acc, out);
}

// Layout of the circuit levels (numerators / denominators per interaction): a fold kernel's lane wants 8 consecutive
// entries (two row pairs), a sums-only kernel's 4, the fraction tree's 2 — 128 / 64 / 32 B at that lane stride. Inside every
// complete block of 512 entries, entry e lives at (e mod 8) * 64 + (e div 8) mod 64: the fold's loads are contiguous 1 KiB
// runs, the sums-only loads two 512 B runs, the tree's four 256 B runs, and the writers (one entry per lane) fill whole
// 128 B lines. The last, incomplete block keeps the natural order (nothing grows; the <= 2-entry level the host reads is
// untouched). Same idea as folded_pos below.
__device__ __forceinline__ uint32_t level_pos(uint32_t e, uint32_t len) {
    return (e | 511u) < len ? ((e & ~511u) | ((e & 7u) << 6) | ((e >> 3) & 63u)) : e;
}

// ================================================================ first layer
// Per (chip, interaction): program words in device memory (layout of sp1_amd/air.py InteractionProgram, one
// interaction: is_send, kind, n_values, vcol(multiplicity), vcol(values..)), column-major traces.
struct IntDesc {
    const uint32_t* prog;      // this interaction's words
    const uint32_t* main;      // column-major [main_w][rows]
    const uint32_t* prep;
    uint32_t rows;
    uint32_t* n_out;           // base numerators [rows]
    Ext* d_out;                // ext denominators [rows]
};

typedef const uint32_t __attribute__((address_space(4)))* prog_words_t;      // interaction programs: wave-uniform, read-only -> scalar loads
__device__ __forceinline__ uint32_t vcol_apply(prog_words_t& p, const IntDesc& d, uint32_t r) {
    const uint32_t nt = p[0];
    uint32_t acc = p[1];                                   // constant (Montgomery)
    p += 2;
    for (uint32_t t = 0; t < nt; t++, p += 3) {
        const uint32_t* col = (p[0] ? d.main : d.prep) + (size_t)p[1] * d.rows;
        const uint32_t v = gptr(col)[r], wgt = p[2];             // most weights are one (a wave-uniform test): no product then
        acc = kb::add(acc, wgt == kb::R1 ? v : kb::mul(v, wgt));
    }
    return acc;
}

// betas: [n_betas] ext (partial Lagrange of beta_seed)
__global__ __launch_bounds__(256) void first_layer_kernel(const IntDesc* __restrict__ descs, Ext alpha, const Ext* __restrict__ betas) {
    const IntDesc d = descs[blockIdx.y];
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < d.rows; r += gridDim.x * blockDim.x) {
        prog_words_t p = (prog_words_t)(uintptr_t)d.prog;
        const bool is_send = p[0] != 0;
        const uint32_t kind = p[1], nv = p[2];            // kind: Montgomery form
        p += 3;
        uint32_t m = vcol_apply(p, d, r);
        if (!is_send) m = kb::sub(0u, m);
        Ext den = kb::ext_add(alpha, kb::ext_mul_base(ld_ext(betas, 0), kind));
        for (uint32_t j = 0; j < nv; j++) den = kb::ext_add(den, kb::ext_mul_base(ld_ext(betas, 1 + j), vcol_apply(p, d, r)));
        const uint32_t rp = level_pos(r, d.rows);
        gptr(d.n_out)[rp] = m;
        st_ext(d.d_out, rp, den);
    }
}

// ================================================================ fraction tree
struct TransDesc {
    const void* n_in;          // base (level L) or ext
    const Ext* d_in;
    Ext* n_out;
    Ext* d_out;
    uint32_t rows_in;
};

template <bool NBASE>
__device__ __forceinline__ Ext load_n(const void* p, uint32_t i) {
    if (NBASE) return kb::ext_from_base(gptr((const uint32_t*)p)[i]);
    return ld_ext((const Ext*)p, i);
}

template <bool NBASE>
__global__ __launch_bounds__(256) void transition_kernel(const TransDesc* __restrict__ descs) {
    const TransDesc d = descs[blockIdx.y];
    const uint32_t rows_out = (d.rows_in + 1) / 2;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows_out; r += gridDim.x * blockDim.x) {
        const uint32_t ia = level_pos(2 * r, d.rows_in), ib = level_pos(2 * r + 1, d.rows_in), io = level_pos(r, rows_out);
        const Ext da = ld_ext(d.d_in, ia);
        if (2 * r + 1 < d.rows_in) {
            const Ext db = ld_ext(d.d_in, ib);
            Ext n;
            if (NBASE) {
                const uint32_t na = gptr((const uint32_t*)d.n_in)[ia], nb = gptr((const uint32_t*)d.n_in)[ib];
                n = kb::ext_add(kb::ext_mul_base(db, na), kb::ext_mul_base(da, nb));
            } else {
                n = kb::ext_add(kb::ext_mul(db, load_n<false>(d.n_in, ia)), kb::ext_mul(da, load_n<false>(d.n_in, ib)));
            }
            st_ext(d.n_out, io, n);
            st_ext(d.d_out, io, kb::ext_mul(da, db));
        } else {                                           // partner is a padding row: (0, 1)
            st_ext(d.n_out, io, load_n<NBASE>(d.n_in, ia));
            st_ext(d.d_out, io, da);
        }
    }
}

// ================================================================ eq tables of one layer
// out holds, for t = 0 .. v, the partial-Lagrange table of the first t coordinates of `pt` at offset 2^t - 1... i.e.
// table t occupies [2^t - 1, 2^(t+1) - 1). Thread g -> (t, i).
struct PointArg { Ext c[32]; };
__global__ __launch_bounds__(256) void eq_prefix_tables_kernel(PointArg pt, int v, Ext* __restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x + 1;       // 1 .. 2^(v+1) - 1
    if (g >= (2u << v)) return;
    const int t = 31 - __clz(g);
    const uint32_t i = g - (1u << t);
    Ext acc = kb::ext_one();
    for (int j = 0; j < t; j++) {
        const bool bit = (i >> (t - 1 - j)) & 1u;
        acc = kb::ext_mul(acc, bit ? pt.c[j] : kb::ext_sub(kb::ext_one(), pt.c[j]));
    }
    st_ext(out, g - 1, acc);
}

// ================================================================ sumcheck rounds over a row variable
// Per (chip, interaction): the sources of one round.
//  FIRST (interleaved): x_n / x_d = level v+1 vectors (rows_x real entries); row r of the layer is the pair
//                       (entry 2r = "0" half, entry 2r+1 = "1" half).
//  later: four separate vectors n0, d0, n1, d1 with `rows` real entries each.
struct RoundDesc {
    const void* src[4];        // FIRST: {x_n, x_d, -, -}; later: {n0, d0, n1, d1}
    Ext* dst[4];               // folded n0, d0, n1, d1
    uint32_t rows;             // real rows r of the layer at this round (FIRST: ceil(rows_x / 2))
    uint32_t rows_x;           // FIRST only
    uint32_t eq_int_index;     // global interaction index
    uint32_t tile0;            // first workgroup of this interaction in the launch (work is split by size, not per interaction)
};

// workgroup -> (interaction, range of its pairs): the interactions differ by orders of magnitude in height, so the
// launch is a flat list of equally sized tiles; descs[].tile0 is ascending
__device__ __forceinline__ uint32_t find_desc(const RoundDesc* __restrict__ descs, uint32_t K, uint32_t b) {
    uint32_t lo = 0, hi = K;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (descs[mid].tile0 <= b) lo = mid; else hi = mid;
    }
    return lo;
}
struct Quad { Ext n0, d0, n1, d1; };

// Layout of the folded tables in scratch: a lane of the next round wants rows 4k .. 4k+3 of each table, i.e. 64 B at a
// 64 B lane stride — every load instruction of a wave would touch 32 cache lines for 1 KiB (measured: the L1/TA rate
// co-bounds the large rounds with the VALUs; with coalesced addresses the same kernel runs 25 % faster). So inside every
// complete block of 256 rows, row r lives at (r mod 4) * 64 + (r div 4) mod 64: the four loads of a wave are four
// contiguous 1 KiB runs, and the stores of the producing round (rows 2k, 2k+1 per lane) are two 512 B runs each. The
// last (incomplete) block of a table keeps the natural order, so tables never grow and short tables (what the host
// fetches after the last fold) are untouched.
__device__ __forceinline__ uint32_t folded_pos(uint32_t r, uint32_t len) {
    return (r | 255u) < len ? ((r & ~255u) | ((r & 3u) << 6) | ((r >> 2) & 63u)) : r;
}

template <bool FIRST, bool NBASE>
__device__ __forceinline__ Quad load_quad(const RoundDesc& d, uint32_t r) {
    Quad q;
    if (r >= d.rows) { q.n0 = q.n1 = kb::ext_zero(); q.d0 = q.d1 = kb::ext_one(); return q; }
    if (FIRST) {
        const uint32_t ea = level_pos(2 * r, d.rows_x), eb = level_pos(2 * r + 1, d.rows_x);
        q.n0 = load_n<NBASE>(d.src[0], ea);
        q.d0 = ld_ext((const Ext*)d.src[1], ea);
        if (2 * r + 1 < d.rows_x) { q.n1 = load_n<NBASE>(d.src[0], eb); q.d1 = ld_ext((const Ext*)d.src[1], eb); }
        else { q.n1 = kb::ext_zero(); q.d1 = kb::ext_one(); }
    } else {
        const uint32_t rp = folded_pos(r, d.rows);
        q.n0 = ld_ext((const Ext*)d.src[0], rp); q.d0 = ld_ext((const Ext*)d.src[1], rp);
        q.n1 = ld_ext((const Ext*)d.src[2], rp); q.d1 = ld_ext((const Ext*)d.src[3], rp);
    }
    return q;
}

__device__ __forceinline__ Ext lerp(const Ext& a, const Ext& b, const Ext& t) { return kb::ext_add(a, kb::ext_mul(kb::ext_sub(b, a), t)); }   // t: the round's challenge (wave-uniform, second)

// accumulate the three sums of one row pair (a = row 2k, b = row 2k+1) weighted by w = eq_int: S0 += w T[2k] F(a),
// Sh += w (T[2k] + T[2k+1]) Fh(a + b), Seq += w (T[2k] + T[2k+1])
__device__ __forceinline__ void accumulate_pair(const Quad& a, const Quad& b, const Ext& lambda, const Ext& ta, const Ext& tb,
                                                Ext (&acc)[3]) {
    const Ext f0 = kb::ext_add(kb::ext_mul(kb::ext_add(kb::ext_mul(a.n0, a.d1), kb::ext_mul(a.n1, a.d0)), lambda), kb::ext_mul(a.d0, a.d1));
    const Ext sn0 = kb::ext_add(a.n0, b.n0), sn1 = kb::ext_add(a.n1, b.n1), sd0 = kb::ext_add(a.d0, b.d0), sd1 = kb::ext_add(a.d1, b.d1);
    const Ext fh = kb::ext_add(kb::ext_mul(kb::ext_add(kb::ext_mul(sn0, sd1), kb::ext_mul(sn1, sd0)), lambda), kb::ext_mul(sd0, sd1));
    const Ext ts = kb::ext_add(ta, tb);
    acc[0] = kb::ext_add(acc[0], kb::ext_mul(ta, f0));
    acc[1] = kb::ext_add(acc[1], kb::ext_mul(ts, fh));
    acc[2] = kb::ext_add(acc[2], ts);
}

// ---------------------------------------------------------------- device-resident transcript for the row rounds
// The v row-variable rounds of a layer used to return to the host after every launch (three sums -> the round polynomial
// -> observe -> sample alpha -> next launch): ~56 us of host round trip around a ~26 us kernel, 231 times per proof. With
// `GkrChainArgs` the LAST workgroup of a round (the one that already reduces the partial sums) finishes the round itself:
// it forms the cubic from the sums (the Lagrange basis of the nodes {0, 1, 1/2, b} only depends on the layer's point:
// the host precomputes it per round), runs the DuplexChallenger on 16 lanes (`permute_coop16`: one state word per lane),
// and leaves alpha / claim / eq factor for the next launch, which the host has already enqueued. The host sees a layer's
// messages, challenges and the sponge once, at the end of the layer.
// (the reference's CUDA path keeps a device challenger for the same reason:
//  /root/reference/sp1-gpu/crates/sys/include/challenger/challenger.cuh:L13-L170)
struct GkrChain {                          // device memory, one per proof
    uint32_t ch_state[16], ch_in[8], ch_out[8];
    uint32_t ch_n_in, ch_n_out, pad0, pad1;
    Ext claim, PA, alpha;
};
struct GkrRoundConst { Ext pt; Ext basis[3][4]; };     // basis[k] = the cubic that is 1 at node k and 0 at the other three
struct GkrRoundOut { Ext poly[4]; Ext alpha; };
struct GkrChainArgs { GkrChain* st; const GkrRoundConst* rc; GkrRoundOut* out; const p2::RoundConstants* p2rc; };

// DuplexChallenger<KoalaBear, 16, 8> spread over the first 16 lanes of a wave (lane r = state word r; lanes 0..7 also
// hold the input / output buffers). Same semantics as the host object in prover.hip.
struct CoopChallenger {
    uint32_t x, in_w, out_w;
    int n_in, n_out;
    uint32_t lane;
    const p2::RoundConstants* rc;
    __device__ __forceinline__ void duplexing() {
        if ((int)lane < n_in) x = in_w;
        n_in = 0;
        x = p2::permute_coop16(x, lane, *rc);
        out_w = x;
        n_out = 8;
    }
    __device__ __forceinline__ void observe(uint32_t v) {      // v is the same in every lane
        n_out = 0;
        if ((int)lane == n_in) in_w = v;
        if (++n_in == 8) duplexing();
    }
    __device__ __forceinline__ uint32_t sample() {
        if (n_in != 0 || n_out == 0) duplexing();
        --n_out;
        return __shfl(out_w, n_out, 16);
    }
};

// Runs in lanes 0..15 of the last workgroup. sums: S0, Sh, Seq (12 words, LDS).
__device__ __forceinline__ void gkr_round_tail(const uint32_t* sums, const GkrChainArgs& ca) {
    const uint32_t lane = threadIdx.x;
    GkrChain* st = ca.st;
    const Ext S0{{sums[0], sums[1], sums[2], sums[3]}}, Sh{{sums[4], sums[5], sums[6], sums[7]}}, Seq{{sums[8], sums[9], sums[10], sums[11]}};
    const Ext claim = ld_ext(&st->claim, 0), PA = ld_ext(&st->PA, 0), pt = ld_ext(&ca.rc->pt, 0);
    const Ext one = kb::ext_one();
    const uint32_t inv8 = kb::inv(kb::to_monty(8u)), four = kb::to_monty(4u);
    const Ext corr = kb::ext_sub(one, Seq);
    const Ext p0 = kb::ext_mul(PA, kb::ext_add(S0, kb::ext_mul(corr, kb::ext_sub(one, pt))));
    const Ext ph = kb::ext_mul_base(kb::ext_mul(PA, kb::ext_add(Sh, kb::ext_mul_base(corr, four))), inv8);
    const Ext ys[3] = {p0, kb::ext_sub(claim, p0), ph};
    Ext poly[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        Ext a = kb::ext_zero();
#pragma unroll
        for (int k = 0; k < 3; k++) a = kb::ext_add(a, kb::ext_mul(ys[k], ld_ext(&ca.rc->basis[k][d], 0)));
        poly[d] = a;
    }
    CoopChallenger ch;
    ch.lane = lane; ch.rc = ca.p2rc;
    ch.x = st->ch_state[lane];
    ch.in_w = lane < 8 ? st->ch_in[lane] : 0u;
    ch.out_w = lane < 8 ? st->ch_out[lane] : 0u;
    ch.n_in = (int)st->ch_n_in; ch.n_out = (int)st->ch_n_out;
#pragma unroll 1
    for (int d = 0; d < 4; d++)
#pragma unroll 1
        for (int k = 0; k < 4; k++) ch.observe(poly[d].c[k]);
    Ext alpha;
#pragma unroll 1
    for (int k = 0; k < 4; k++) alpha.c[k] = ch.sample();
    const Ext next_claim = kb::ext_add(kb::ext_mul(kb::ext_add(kb::ext_mul(kb::ext_add(kb::ext_mul(poly[3], alpha), poly[2]), alpha), poly[1]), alpha), poly[0]);
    const Ext next_PA = kb::ext_mul(PA, kb::ext_add(kb::ext_mul(pt, alpha), kb::ext_mul(kb::ext_sub(one, pt), kb::ext_sub(one, alpha))));
    st->ch_state[lane] = ch.x;
    if (lane < 8) { st->ch_in[lane] = ch.in_w; st->ch_out[lane] = ch.out_w; }
    if (lane == 0) {
        st->ch_n_in = (uint32_t)ch.n_in; st->ch_n_out = (uint32_t)ch.n_out;
        st_ext(&st->claim, 0, next_claim); st_ext(&st->PA, 0, next_PA); st_ext(&st->alpha, 0, alpha);
#pragma unroll
        for (int d = 0; d < 4; d++) st_ext(&ca.out->poly[d], 0, poly[d]);
        st_ext(&ca.out->alpha, 0, alpha);
    }
}

// rs_finish (round_sync.hpp) with the round finished on the device instead of published to the host
template <int NS>
__device__ __forceinline__ void rs_finish_chain(const Ext (&acc)[NS], uint32_t* __restrict__ partials, uint32_t block_linear,
                                                uint32_t total_blocks, uint32_t* counter, const GkrChainArgs& ca) {
    static_assert(NS == 3, "the GKR rounds publish three sums");
    __shared__ uint32_t sm[4][4 * NS];
    __shared__ uint32_t fin[4 * NS];
    __shared__ uint32_t last_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = wave_sum(acc[s].c[k]);
            if (lane == 0) sm[wave][4 * s + k] = w;
        }
    __syncthreads();
    if (threadIdx.x < 4 * NS) {
        uint32_t a = 0;
        for (int i = 0; i < 4; i++) a = kb::add(a, sm[i][threadIdx.x]);
        rs_store_partial(&partials[(size_t)block_linear * 4 * NS + threadIdx.x], a);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_flag = rs_ticket_is_last(counter, block_linear, total_blocks);
    __syncthreads();
    if (!last_flag) return;
    Ext tot[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) tot[s] = kb::ext_zero();
    for (uint32_t i = threadIdx.x; i < total_blocks; i += 256)
#pragma unroll
        for (int s = 0; s < NS; s++) tot[s] = kb::ext_add(tot[s], rs_load_partial(partials + ((size_t)i * NS + s) * 4));
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = wave_sum(tot[s].c[k]);
            if (lane == 0) sm[wave][4 * s + k] = w;
        }
    __syncthreads();
    if (threadIdx.x < 4 * NS) {
        uint32_t a = 0;
        for (int i = 0; i < 4; i++) a = kb::add(a, sm[i][threadIdx.x]);
        fin[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < 16) gkr_round_tail(fin, ca);
}

// Small rounds (FLAT): most of a proof's ~230 row rounds have a handful of pairs per interaction — one workgroup per
// interaction is then 730 workgroups of one busy lane each, and what the launch costs is the 730 tickets and partial sums of
// its tail. FLAT launches give every LANE one pair instead: lane p of the launch looks its descriptor up in a host-built
// table (flat_index[p]; descs[].tile0 = first pair of the interaction), weighs its own sums with its interaction's eq
// factor, and the launch is total_pairs / 256 workgroups.
struct FlatArgs { const uint16_t* index; uint32_t total_pairs; };

// round 0 of a layer: sums only. T = partial-Lagrange table of the layer's row point (2^v entries)
template <bool NBASE, bool FLAT>
__global__ __launch_bounds__(256) void round_sum_first(const RoundDesc* __restrict__ descs, const Ext* __restrict__ eq_int,
                                                       const Ext* __restrict__ T, Ext lambda, uint32_t* __restrict__ partials,
                                                       RoundSync rs, uint32_t seq, uint32_t K, uint32_t tile_size, GkrChainArgs ca, FlatArgs fa) {
    Ext acc[3] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
    if (FLAT) {
        const uint32_t p = blockIdx.x * 256 + threadIdx.x;
        if (p < fa.total_pairs) {
            const RoundDesc d = descs[fa.index[p]];
            const uint32_t k = p - d.tile0;
            const Quad a = load_quad<true, NBASE>(d, 2 * k), b = load_quad<true, NBASE>(d, 2 * k + 1);
            accumulate_pair(a, b, lambda, ld_ext(T, 2 * k), ld_ext(T, 2 * k + 1), acc);
            const Ext w = ld_ext(eq_int, d.eq_int_index);
#pragma unroll
            for (int s = 0; s < 3; s++) acc[s] = kb::ext_mul(acc[s], w);
        }
    } else {
        const RoundDesc d = descs[find_desc(descs, K, blockIdx.x)];
        const uint32_t pairs = (d.rows + 1) / 2;
        const uint32_t k0 = (blockIdx.x - d.tile0) * tile_size, k1 = min(pairs, k0 + tile_size);
        for (uint32_t k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
            const Quad a = load_quad<true, NBASE>(d, 2 * k), b = load_quad<true, NBASE>(d, 2 * k + 1);
            accumulate_pair(a, b, lambda, ld_ext(T, 2 * k), ld_ext(T, 2 * k + 1), acc);
        }
        const Ext w = ld_ext(eq_int, d.eq_int_index);
#pragma unroll
        for (int s = 0; s < 3; s++) acc[s] = kb::ext_mul(acc[s], w);
    }
    if (ca.st) rs_finish_chain<3>(acc, partials, blockIdx.x, gridDim.x, rs.counter, ca);
    else rs_finish<3>(acc, partials, blockIdx.x, gridDim.x, rs, seq);
}

// rows (4k .. 4k+3) -> folded rows (2k, 2k+1), stored; and (if SUM) the next round's sums from the folded pair.
// Split in two so that the tiled loop can issue the loads of iteration i + 1 before the arithmetic of iteration i.
// Software-pipelined variant of the tiled loop (loads of iteration i + 1 in flight during the arithmetic of iteration i):
// 182 VGPRs -> 2 waves per SIMD instead of 4. Measured on the core-shaped shard: later folds -4 %, a layer's first fold -9 %,
// the top layer's first fold (base-field numerators) +7 %: no net gain — with the lane-blocked layouts the large rounds
// are bound by VALU issue (2,400 instructions per iteration, a third of them half-rate 64-bit multiply-adds: ~87 % of that
// bound), not by exposed latency. Off; kept for A/B runs.
constexpr bool PREFETCH_FOLDS = false;
struct FoldIn { Quad in[4]; Ext ta, tb; };
template <bool FIRST, bool NBASE, bool SUM>
__device__ __forceinline__ void fold_load(const RoundDesc& d, uint32_t k, const Ext* __restrict__ T_next, FoldIn& f) {
    // every load of the iteration is issued before the first use: one exposed memory latency per iteration instead
    // of three (rows of h = 0, rows of h = 1, eq table)
#pragma unroll
    for (int q = 0; q < 4; q++) f.in[q] = load_quad<FIRST, NBASE>(d, 4 * k + q);
    if (SUM) { f.ta = ld_ext(T_next, 2 * k); f.tb = ld_ext(T_next, 2 * k + 1); }
}
template <bool SUM>
__device__ __forceinline__ void fold_compute(const RoundDesc& d, uint32_t k, uint32_t rows_out, const Ext& alpha, const Ext& lambda,
                                             const FoldIn& f, Ext (&acc)[3]) {
    Quad o[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t ro = 2 * k + h;
        const Quad& a = f.in[2 * h];
        const Quad& b = f.in[2 * h + 1];
        o[h].n0 = lerp(a.n0, b.n0, alpha); o[h].d0 = lerp(a.d0, b.d0, alpha);
        o[h].n1 = lerp(a.n1, b.n1, alpha); o[h].d1 = lerp(a.d1, b.d1, alpha);
        // (the fold that binds the last row variable, SUM = false, leaves one row per table for the host: natural order)
        const uint32_t rq = SUM ? folded_pos(ro, rows_out) : ro;
        if (ro < rows_out) { st_ext(d.dst[0], rq, o[h].n0); st_ext(d.dst[1], rq, o[h].d0); st_ext(d.dst[2], rq, o[h].n1); st_ext(d.dst[3], rq, o[h].d1); }
    }
    if (SUM) accumulate_pair(o[0], o[1], lambda, f.ta, f.tb, acc);
}
template <bool FIRST, bool NBASE, bool SUM>
__device__ __forceinline__ void fold_sum_pair(const RoundDesc& d, uint32_t k, uint32_t rows_out, const Ext& alpha, const Ext& lambda,
                                              const Ext* __restrict__ T_next, Ext (&acc)[3]) {
    FoldIn f;
    fold_load<FIRST, NBASE, SUM>(d, k, T_next, f);
    fold_compute<SUM>(d, k, rows_out, alpha, lambda, f, acc);
}

// fold rows (2r', 2r'+1) -> r' with alpha for r' = 2k, 2k+1, store, and (if SUM) accumulate the next round's sums
// from the folded pair. T_next = table of the remaining row variables (half the size).
template <bool FIRST, bool NBASE, bool SUM, bool FLAT>
__global__ __launch_bounds__(256) void round_fold_sum(const RoundDesc* __restrict__ descs, const Ext* __restrict__ eq_int,
                                                      const Ext* __restrict__ T_next, Ext lambda, Ext alpha_arg,
                                                      uint32_t* __restrict__ partials, RoundSync rs, uint32_t seq, uint32_t K,
                                                      uint32_t tile_size, GkrChainArgs ca, FlatArgs fa) {
    const Ext alpha = ca.st ? ld_ext(&ca.st->alpha, 0) : alpha_arg;      // chained: left by the previous round's last workgroup
    Ext acc[3] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
    if (FLAT) {
        const uint32_t p = blockIdx.x * 256 + threadIdx.x;
        if (p < fa.total_pairs) {
            const RoundDesc d = descs[fa.index[p]];
            fold_sum_pair<FIRST, NBASE, SUM>(d, p - d.tile0, (d.rows + 1) / 2, alpha, lambda, T_next, acc);
            if (SUM) {
                const Ext w = ld_ext(eq_int, d.eq_int_index);
#pragma unroll
                for (int s = 0; s < 3; s++) acc[s] = kb::ext_mul(acc[s], w);
            }
        }
    } else {
        const RoundDesc d = descs[find_desc(descs, K, blockIdx.x)];
        const uint32_t rows_out = (d.rows + 1) / 2;
        const uint32_t pairs = (rows_out + 1) / 2;
        const uint32_t k0 = (blockIdx.x - d.tile0) * tile_size, k1 = min(pairs, k0 + tile_size);
        if (PREFETCH_FOLDS && SUM) {
            // software pipeline: the loads of iteration i + 1 are in flight during the arithmetic of iteration i
            uint32_t k = k0 + threadIdx.x;
            FoldIn cur;
            if (k < k1) fold_load<FIRST, NBASE, SUM>(d, k, T_next, cur);
            while (k < k1) {
                const uint32_t kn = k + blockDim.x;
                FoldIn nxt;
                if (kn < k1) fold_load<FIRST, NBASE, SUM>(d, kn, T_next, nxt);
                fold_compute<SUM>(d, k, rows_out, alpha, lambda, cur, acc);
                if (kn < k1) cur = nxt;
                k = kn;
            }
        } else {
            for (uint32_t k = k0 + threadIdx.x; k < k1; k += blockDim.x) fold_sum_pair<FIRST, NBASE, SUM>(d, k, rows_out, alpha, lambda, T_next, acc);
        }
        if (SUM) {
            const Ext w = ld_ext(eq_int, d.eq_int_index);
#pragma unroll
            for (int s = 0; s < 3; s++) acc[s] = kb::ext_mul(acc[s], w);
        }
    }
    if (SUM) {
        if (ca.st) rs_finish_chain<3>(acc, partials, blockIdx.x, gridDim.x, rs.counter, ca);
        else rs_finish<3>(acc, partials, blockIdx.x, gridDim.x, rs, seq);
    }
}

// ================================================================ trace openings at the final point
struct OpenDesc { const uint32_t* cols; uint32_t rows, width, col0, out0; };   // one group of <= OPEN_COLS columns of one chip
constexpr int OPEN_COLS = 4, OPEN_ROWS = 16384;
// Column sums against eq with delayed reduction (kb::DotAcc): 64 terms per lane, one reduction per column and lane.
__global__ __launch_bounds__(256) void open_columns_kernel(const OpenDesc* __restrict__ descs, const uint32_t* __restrict__ eq,
                                                           uint32_t eq_len, uint32_t* __restrict__ partials, uint32_t total_cols) {
    // consecutive workgroups = the column groups of one table over the SAME rows: their eq slice is re-read from the
    // caches, not from HBM
    const OpenDesc d = descs[blockIdx.x];
    const uint32_t r0 = blockIdx.y * OPEN_ROWS;
    if (r0 >= d.rows) return;                             // partials are zero-initialised
    kb::DotAcc acc[OPEN_COLS];
#pragma unroll
    for (int c = 0; c < OPEN_COLS; c++) kb::dot_init(acc[c]);
    for (uint32_t r = r0 + threadIdx.x; r < r0 + OPEN_ROWS && r < d.rows; r += 256) {
        const Ext e{{eq[r], eq[eq_len + r], eq[2 * (size_t)eq_len + r], eq[3 * (size_t)eq_len + r]}};
#pragma unroll
        for (int c = 0; c < OPEN_COLS; c++)
            if (d.col0 + c < d.width) kb::dot_add(acc[c], e, gptr(d.cols)[(size_t)(d.col0 + c) * d.rows + r]);
    }
    __shared__ uint32_t sm[4][4 * OPEN_COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < OPEN_COLS; c++) {
        const Ext v = kb::dot_finish(acc[c]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = wave_sum(v.c[k]);
            if (lane == 0) sm[wave][4 * c + k] = w;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4 * OPEN_COLS) {
        const int c = threadIdx.x / 4;
        if (d.col0 + c < d.width) {
            uint32_t a = 0;
            for (int i = 0; i < 4; i++) a = kb::add(a, sm[i][threadIdx.x]);
            partials[((size_t)blockIdx.y * total_cols + d.out0 + c) * 4 + (threadIdx.x & 3)] = a;
        }
    }
}
__global__ void open_sum_kernel(const uint32_t* __restrict__ partials, uint32_t n_chunks, uint32_t n_words, uint32_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_words) return;
    uint32_t acc = 0;
    for (uint32_t c = 0; c < n_chunks; c++) acc = kb::add(acc, partials[(size_t)c * n_words + j]);
    out[j] = acc;
}

// ================================================================ host side
Ext operator+(const Ext& a, const Ext& b) { return kb::ext_add(a, b); }
Ext operator-(const Ext& a, const Ext& b) { return kb::ext_sub(a, b); }
Ext operator*(const Ext& a, const Ext& b) { return kb::ext_mul(a, b); }
Ext ext_c(uint32_t canonical) { return kb::ext_from_base(kb::to_monty(canonical)); }
int log2_ceil(uint64_t x) { int l = 0; while (((uint64_t)1 << l) < x) l++; return l; }

std::vector<Ext> partial_lagrange_host(const std::vector<Ext>& pt) {
    std::vector<Ext> ev{kb::ext_one()};
    for (const Ext& x : pt) {
        std::vector<Ext> nx(ev.size() * 2);
        for (size_t i = 0; i < ev.size(); i++) { const Ext pr = ev[i] * x; nx[2 * i] = ev[i] - pr; nx[2 * i + 1] = pr; }
        ev.swap(nx);
    }
    return ev;
}
Ext eval_mle_host(const std::vector<Ext>& vals, const std::vector<Ext>& pt) {
    const std::vector<Ext> eq = partial_lagrange_host(pt);
    Ext acc = kb::ext_zero();
    for (size_t i = 0; i < vals.size(); i++) acc = acc + eq[i] * vals[i];
    return acc;
}

using Poly4 = std::array<Ext, 4>;
Ext poly_eval(const Poly4& c, const Ext& x) { return ((c[3] * x + c[2]) * x + c[1]) * x + c[0]; }

// the cubic through (0, y0), (1, y1), (1/2, yh), (b, 0): Lagrange interpolation as the reference's
// interpolate_univariate_polynomial (univariate.rs:L85-L97) — any exact method gives the same coefficients
Poly4 interpolate4(const Ext (&xs)[4], const Ext (&ys)[4]) {
    Poly4 res{kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
    // a node whose value is zero contributes nothing (every caller's fourth value is the root of the eq factor), and
    // the remaining denominators are inverted together (one inversion + 3 products per extra denominator): this runs
    // once per sumcheck round on the host, between two device hand-overs
    Ext num[4][4], den[4];
    bool live[4];
    for (int i = 0; i < 4; i++) {
        live[i] = !kb::ext_eq(ys[i], kb::ext_zero());
        if (!live[i]) continue;
        den[i] = kb::ext_one();
        Ext cur[4] = {ys[i], kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
        int deg = 0;
        for (int j = 0; j < 4; j++) {
            if (j == i) continue;
            den[i] = den[i] * (xs[i] - xs[j]);
            Ext nxt[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
            for (int k = 0; k <= deg; k++) { nxt[k + 1] = nxt[k + 1] + cur[k]; nxt[k] = nxt[k] - cur[k] * xs[j]; }
            deg++;
            for (int k = 0; k < 4; k++) cur[k] = nxt[k];
        }
        for (int k = 0; k < 4; k++) num[i][k] = cur[k];
    }
    // batch inversion of the live denominators
    Ext prefix[4], running = kb::ext_one();
    for (int i = 0; i < 4; i++) if (live[i]) { prefix[i] = running; running = running * den[i]; }
    Ext inv_all = kb::ext_inv(running);
    for (int i = 3; i >= 0; i--) {
        if (!live[i]) continue;
        const Ext inv = inv_all * prefix[i];
        inv_all = inv_all * den[i];
        for (int k = 0; k < 4; k++) res[k] = res[k] + num[i][k] * inv;
    }
    return res;
}

struct Bytes {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void felt(uint32_t m) { const uint32_t v = kb::from_monty(m); for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void ext(const Ext& e) { for (int k = 0; k < 4; k++) felt(e.c[k]); }
};

void observe_ext(sp1hip_challenger_t* ch, const Ext& e) { for (int k = 0; k < 4; k++) challenger_observe(ch, e.c[k]); }

int upload(DeviceBuf& buf, const void* src, size_t bytes, hipStream_t s, PinnedStage& stage) {
    SP1HIP_TRY(buf.alloc(std::max<size_t>(bytes, 16), s));
    return stage.upload(buf.p, src, bytes);
}

struct ChipInfo {
    std::string name;
    uint32_t rows, k, main_w, prep_w, int0;                 // k interactions, int0 = first global interaction index
    const uint32_t* d_main;
    const uint32_t* d_prep;
};

constexpr uint32_t MAX_TILES = 512;
uint32_t tiles_for(uint64_t threads) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>((threads + 255) / 256, 1), MAX_TILES); }
}  // namespace gkr
}  // namespace sp1hip

using namespace sp1hip;
using namespace sp1hip::gkr;

extern "C" {

int sp1hip_logup_gkr_prove(const sp1hip_gkr_chip_t* chips, int n_chips, int max_log_row_count, sp1hip_challenger_t* challenger,
                           uint8_t* h_proof, size_t* proof_len, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(chips && n_chips > 0 && challenger && proof_len, "bad argument");
    const int L = max_log_row_count;
    SP1HIP_REQUIRE(L >= 1 && L <= 30, "max_log_row_count out of range");
    hipStream_t s = S(stream);
    // SP1HIP_GKR_DEBUG=1: host-side phase times on stderr (where a stage's non-kernel time goes)
    static const bool gkr_debug = getenv("SP1HIP_GKR_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!gkr_debug) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sp1hip gkr] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    // ---- parse the interaction programs (host words -> Montgomery device words), gather shapes
    std::vector<ChipInfo> info(n_chips);
    std::vector<std::vector<uint32_t>> progs;               // per global interaction: device-form words
    size_t max_arity = 0, total_cols = 0;
    for (int c = 0; c < n_chips; c++) {
        const sp1hip_gkr_chip_t& ci = chips[c];
        SP1HIP_REQUIRE(ci.name && ci.interactions, "null chip field");
        SP1HIP_REQUIRE(ci.real_rows <= ((uint64_t)1 << L), "chip taller than 2^max_log_row_count");
        SP1HIP_REQUIRE(ci.real_rows == 0 || ci.main_width == 0 || ci.d_main, "null main trace");
        SP1HIP_REQUIRE(ci.real_rows == 0 || ci.prep_width == 0 || ci.d_prep, "null preprocessed trace");
        if (c) SP1HIP_REQUIRE(strcmp(chips[c - 1].name, ci.name) < 0, "chips must be sorted by name (BTreeSet order)");
        info[c] = ChipInfo{ci.name, (uint32_t)ci.real_rows, 0, ci.main_width, ci.prep_width, (uint32_t)progs.size(), ci.d_main, ci.d_prep};
        const uint32_t* p = ci.interactions;
        const uint32_t* end = p + ci.n_words;
        SP1HIP_REQUIRE(ci.n_words >= 1, "empty interaction program");
        const uint32_t ni = *p++;
        info[c].k = ni;
        auto vcol = [&](std::vector<uint32_t>& out) -> bool {
            if (p + 2 > end) return false;
            const uint32_t nt = p[0];
            if (p[1] >= kb::P || p + 2 + 3 * (size_t)nt > end) return false;
            out.push_back(nt);
            out.push_back(kb::to_monty(p[1]));
            p += 2;
            for (uint32_t t = 0; t < nt; t++, p += 3) {
                if (p[0] > 1 || p[2] >= kb::P) return false;
                if (p[1] >= (p[0] ? ci.main_width : ci.prep_width)) return false;
                out.push_back(p[0]); out.push_back(p[1]); out.push_back(kb::to_monty(p[2]));
            }
            return true;
        };
        for (uint32_t i = 0; i < ni; i++) {
            SP1HIP_REQUIRE(p + 3 <= end, "truncated interaction program");
            std::vector<uint32_t> w{p[0], kb::to_monty(p[1] % kb::P), p[2]};
            const uint32_t nv = p[2];
            SP1HIP_REQUIRE(p[0] <= 1 && p[1] < kb::P && nv < 64, "bad interaction header");
            p += 3;
            SP1HIP_REQUIRE(vcol(w), "bad multiplicity column");
            for (uint32_t j = 0; j < nv; j++) SP1HIP_REQUIRE(vcol(w), "bad value column (index or weight out of range)");
            max_arity = std::max<size_t>(max_arity, nv + 1);
            progs.push_back(std::move(w));
        }
        SP1HIP_REQUIRE(p == end, "trailing words in interaction program");
        total_cols += (size_t)ci.main_width + ci.prep_width;
    }
    const uint32_t K = (uint32_t)progs.size();
    SP1HIP_REQUIRE(K >= 1, "no interactions in the shard");
    const int niv = log2_ceil(K), beta_seed_dim = log2_ceil(max_arity);
    const uint32_t W = 1u << niv;

    // ---- proof size (everything is determined by the shapes)
    size_t need = 2 * (8 + (size_t)2 * W * 16 + 24) + 8;
    for (int v = 1; v <= L - 1; v++) need += 64 + 8 + (size_t)(niv + v) * (8 + 64) + 16 + 8 + (size_t)(niv + v) * 16 + 16;
    need += 8 + (size_t)L * 16 + 8;
    for (int c = 0; c < n_chips; c++) {
        need += 8 + info[c].name.size() + 8 + (size_t)info[c].main_w * 16 + 16 + 1;
        if (info[c].prep_w) need += 8 + (size_t)info[c].prep_w * 16 + 16;
    }
    need += 4;
    if (!h_proof || *proof_len < need) {
        *proof_len = need;
        set_error("sp1hip_logup_gkr_prove: proof buffer too small, need %zu bytes", need);
        return SP1HIP_ERROR_BUFFER_TOO_SMALL;
    }

    const DeviceCtx* ctx;                                    // (validation and the size query above need no device)
    SP1HIP_TRY(get_device_ctx(&ctx));
    sp1hip_challenger_t* ch = nullptr;
    SP1HIP_TRY(sp1hip_challenger_clone(challenger, &ch));
    struct ChGuard { sp1hip_challenger_t* c; ~ChGuard() { sp1hip_challenger_free(c); } } guard{ch};

    // ---- transcript head (prover.rs:L86-L95)
    uint32_t witness = 0;
    SP1HIP_TRY(sp1hip_challenger_grind(ch, 12, &witness, stream));
    const Ext alpha = challenger_sample_ext(ch);
    std::vector<Ext> beta_seed(beta_seed_dim);
    for (auto& b : beta_seed) b = challenger_sample_ext(ch);
    (void)challenger_sample_ext(ch);                         // _pv_challenge
    const std::vector<Ext> betas = partial_lagrange_host(beta_seed);

    // ---- device storage: levels L .. 1 of the fraction tree, per interaction
    // rows at level l of chip c: ceil(h / 2^(L - l))
    auto rows_at = [&](uint32_t h, int l) -> uint32_t { return (uint32_t)(((uint64_t)h + (((uint64_t)1 << (L - l)) - 1)) >> (L - l)); };
    std::vector<uint32_t> int_chip(K);
    for (int c = 0; c < n_chips; c++) for (uint32_t i = 0; i < info[c].k; i++) int_chip[info[c].int0 + i] = c;
    std::vector<size_t> level_entries(L + 2, 0);            // total entries of level l over all interactions
    for (int l = 1; l <= L; l++) for (uint32_t i = 0; i < K; i++) level_entries[l] += rows_at(info[int_chip[i]].rows, l);
    std::vector<DeviceBuf> lvN(L + 1), lvD(L + 1);
    for (int l = 1; l <= L; l++) {
        SP1HIP_TRY(lvN[l].alloc(std::max<size_t>(level_entries[l], 1) * (l == L ? 4 : 16), s));
        SP1HIP_TRY(lvD[l].alloc(std::max<size_t>(level_entries[l], 1) * 16, s));
    }
    // offsets of interaction i inside level l
    std::vector<std::vector<size_t>> off(L + 1, std::vector<size_t>(K + 1, 0));
    for (int l = 1; l <= L; l++) for (uint32_t i = 0; i < K; i++) off[l][i + 1] = off[l][i] + rows_at(info[int_chip[i]].rows, l);
    auto n_ptr = [&](int l, uint32_t i) -> void* { return (char*)lvN[l].p + off[l][i] * (l == L ? 4 : 16); };
    auto d_ptr = [&](int l, uint32_t i) -> Ext* { return lvD[l].ext() + off[l][i]; };

    mark("parse + level buffers");
    // ---- first layer
    PinnedStage stage;                                       // small uploads (round_sync.hpp)
    SP1HIP_TRY(stage.init(s));
    DeviceBuf d_progs, d_betas, d_descs;
    uint32_t max_rows = 0;
    std::vector<uint32_t> flat;                              // upload sources live to the end of the call
    std::vector<IntDesc> descs(K);
    {
        std::vector<size_t> poff(K);
        for (uint32_t i = 0; i < K; i++) { poff[i] = flat.size(); flat.insert(flat.end(), progs[i].begin(), progs[i].end()); }
        SP1HIP_TRY(upload(d_progs, flat.data(), flat.size() * 4, s, stage));
        SP1HIP_TRY(upload(d_betas, betas.data(), betas.size() * 16, s, stage));
        for (uint32_t i = 0; i < K; i++) {
            const ChipInfo& c = info[int_chip[i]];
            descs[i] = IntDesc{d_progs.u32() + poff[i], c.d_main, c.d_prep, c.rows, (uint32_t*)n_ptr(L, i), d_ptr(L, i)};
            max_rows = std::max(max_rows, c.rows);
        }
        SP1HIP_TRY(upload(d_descs, descs.data(), descs.size() * sizeof(IntDesc), s, stage));
        if (max_rows) {
            ScopedTimer t("gkr_first_layer", s);
            hipLaunchKernelGGL(first_layer_kernel, dim3(tiles_for(max_rows), K), dim3(256), 0, s, (const IntDesc*)d_descs.p, alpha,
                               (const Ext*)d_betas.p);
            SP1HIP_LAUNCH_CHECK();
        }
    }
    mark("first layer enqueued");
    // ---- fraction tree
    // every level's descriptors are planned and uploaded once: the tree is built by L - 1 back-to-back launches
    DeviceBuf d_trans;
    std::vector<TransDesc> tdesc((size_t)K * (L >= 2 ? L - 1 : 0));
    std::vector<uint32_t> level_mr(L + 1, 0);
    for (int l = L; l >= 2; l--)
        for (uint32_t i = 0; i < K; i++) {
            const uint32_t rin = rows_at(info[int_chip[i]].rows, l);
            tdesc[(size_t)(L - l) * K + i] = TransDesc{n_ptr(l, i), d_ptr(l, i), (Ext*)n_ptr(l - 1, i), d_ptr(l - 1, i), rin};
            level_mr[l] = std::max(level_mr[l], (rin + 1) / 2);
        }
    SP1HIP_TRY(upload(d_trans, tdesc.data(), tdesc.size() * sizeof(TransDesc), s, stage));
    for (int l = L; l >= 2; l--) {
        const uint32_t mr = level_mr[l];
        if (!mr) continue;
        const TransDesc* d_t = (const TransDesc*)d_trans.p + (size_t)(L - l) * K;
        ScopedTimer t("gkr_transition", s);
        if (l == L) hipLaunchKernelGGL(transition_kernel<true>, dim3(tiles_for(mr), K), dim3(256), 0, s, d_t);
        else hipLaunchKernelGGL(transition_kernel<false>, dim3(tiles_for(mr), K), dim3(256), 0, s, d_t);
        SP1HIP_LAUNCH_CHECK();
    }
    mark("tree enqueued");
    Mailbox mb;                                              // device -> host hand-overs outside the sumcheck rounds
    SP1HIP_TRY(mb.init(s));
    // (the descriptors of every round depend on shapes only: they are planned and uploaded WHILE the GPU builds the first
    // layer and the fraction tree — 3.9 ms of host work at 730 interactions x 231 rounds that used to sit on the critical
    // path behind the first hand-over)
    // ---- GKR rounds, layer v = 1 .. L-1 (reads level v + 1)
    struct RoundOut { Ext n0, n1, d0, d1; std::vector<Poly4> polys; Ext claimed_sum, eval; std::vector<Ext> point; };
    std::vector<RoundOut> rounds;
    DeviceBuf d_eq_int, d_T, d_partials, d_out, scratch[2];
    SP1HIP_TRY(d_eq_int.alloc((size_t)W * 16, s));
    SP1HIP_TRY(d_T.alloc(((size_t)2 << std::max(L - 1, 1)) * 16, s));
    SP1HIP_TRY(d_out.alloc(48, s));
    // folded tables: 4 vectors per interaction, at most ceil(rows(level v+1) / 4) entries each after the first fold
    size_t scratch_entries = 0;
    for (uint32_t i = 0; i < K; i++) scratch_entries += (rows_at(info[int_chip[i]].rows, L) + 3) / 4 + 1;
    for (int b = 0; b < 2; b++) SP1HIP_TRY(scratch[b].alloc(std::max<size_t>(scratch_entries, 1) * 64, s));
    // The descriptors of EVERY round of EVERY layer depend on shapes only: plan them all, upload once, and let each
    // launch index the table — no per-round host-to-device copy on the critical path.
    auto scratch_ptr = [&](int b, uint32_t i, int which, const std::vector<size_t>& so) -> Ext* { return scratch[b].ext() + 4 * so[i] + (size_t)which * (so[i + 1] - so[i]); };
    // fills K descriptors of one launch. j = round index inside the layer (0 = sums only, >= 1 fold of round j-1),
    // last = the fold that binds the last row variable
    struct LaunchShape { uint32_t tiles, tile_size, total_pairs; size_t flat_off; bool flat; };
    std::vector<LaunchShape> shapes;
    std::vector<uint16_t> flat_index;                        // FLAT launches: pair -> descriptor, all launches back to back
    const uint32_t FLAT_MAX_PAIRS = [] { const char* e = getenv("SP1HIP_GKR_FLAT_PAIRS"); return e ? (uint32_t)atoi(e) : 65536u; }();   // read per call (tests)
    // workgroups of a large round. While every workgroup paid an L2 write-back and a serialised ticket in its tail
    // (round_sync.hpp) one resident set — 256 CUs x 4 workgroups — was the optimum; without them finer tiles balance the
    // tail better. GKR kernels on the core-shaped shard, ms (SP1HIP_GKR_TILES): 512: 24.6, 768: 22.3, 1024: 21.4,
    // 2048: 21.0, 3072: 20.8, 4096: 20.6, 6144: 20.8, 8192: 20.9, 12288: 21.2.
    static const uint32_t TARGET_TILES = [] { const char* e = getenv("SP1HIP_GKR_TILES"); return e ? std::max<uint32_t>((uint32_t)atoi(e), 1u) : 4096u; }();
    auto fill_descs = [&](RoundDesc* out, int v, int j, bool last, const std::vector<uint32_t>& live, int cur,
                          const std::vector<size_t>& so_prev, const std::vector<size_t>& so_next) {
        // pairs handled per interaction: sums-only launch: ceil(rows / 2); fold launches: ceil(ceil(rows / 2) / 2)
        uint64_t total_pairs = 0;
        auto pairs_of = [&](uint32_t rows) -> uint32_t { const uint32_t p = (rows + 1) / 2; return j == 0 ? p : (p + 1) / 2; };
        for (uint32_t i = 0; i < K; i++) total_pairs += pairs_of(live[i]);
        const bool flat = total_pairs <= FLAT_MAX_PAIRS && K <= 65536 && !last;
        const uint32_t tile_size = flat ? 1u : (uint32_t)std::max<uint64_t>(256, ((total_pairs + TARGET_TILES - 1) / TARGET_TILES + 255) / 256 * 256);
        const size_t flat_off = flat_index.size();
        uint32_t tile0 = 0;
        for (uint32_t i = 0; i < K; i++) {
            RoundDesc& d = out[i];
            d = RoundDesc{};
            d.rows = live[i]; d.eq_int_index = i;
            const bool from_level = j == 0 || (last ? v == 1 : j == 1);
            if (from_level) { d.src[0] = n_ptr(v + 1, i); d.src[1] = d_ptr(v + 1, i); d.rows_x = rows_at(info[int_chip[i]].rows, v + 1); }
            else for (int w = 0; w < 4; w++) d.src[w] = scratch_ptr(cur ^ 1, i, w, so_prev);
            if (j > 0 || last) for (int w = 0; w < 4; w++) d.dst[w] = scratch_ptr(cur, i, w, so_next);
            d.tile0 = tile0;
            const uint32_t n_tiles = (pairs_of(live[i]) + tile_size - 1) / tile_size;
            if (flat) flat_index.insert(flat_index.end(), n_tiles, (uint16_t)i);
            tile0 += n_tiles;
        }
        if (flat) shapes.push_back(LaunchShape{std::max<uint32_t>((tile0 + 255) / 256, 1), tile_size, tile0, flat_off, true});
        else shapes.push_back(LaunchShape{std::max<uint32_t>(tile0, 1), tile_size, tile0, 0, false});
    };
    std::vector<RoundDesc> all_descs;
    for (int v = 1; v <= L - 1; v++) {
        std::vector<uint32_t> live(K);
        for (uint32_t i = 0; i < K; i++) live[i] = (rows_at(info[int_chip[i]].rows, v + 1) + 1) / 2;
        int cur = 0;
        std::vector<size_t> so_prev, so_next;
        for (int j = 0; j <= v; j++) {
            const bool last = j == v;
            if (j > 0) { so_next.assign(K + 1, 0); for (uint32_t i = 0; i < K; i++) so_next[i + 1] = so_next[i] + (live[i] + 1) / 2; }
            all_descs.resize(all_descs.size() + K);
            fill_descs(all_descs.data() + all_descs.size() - K, v, j, last, live, cur, so_prev, so_next);
            if (j > 0 && !last) { for (uint32_t i = 0; i < K; i++) live[i] = (live[i] + 1) / 2; so_prev = so_next; cur ^= 1; }
        }
    }
    {   // partial sums: one slot per workgroup of the largest launch
        uint32_t max_tiles = 1;
        for (auto& sh : shapes) max_tiles = std::max(max_tiles, sh.tiles);
        SP1HIP_TRY(d_partials.alloc((size_t)max_tiles * 48, s));
    }
    mark("round descriptors planned");
    DeviceBuf d_all;
    SP1HIP_TRY(upload(d_all, all_descs.data(), all_descs.size() * sizeof(RoundDesc), s, stage));     // all_descs outlives the copy
    DeviceBuf d_flat;
    if (flat_index.empty()) flat_index.push_back(0);
    flat_index.resize((flat_index.size() + 1) / 2 * 2);
    SP1HIP_TRY(upload(d_flat, flat_index.data(), flat_index.size() * sizeof(uint16_t), s, stage));
    mark("round descriptors uploaded");
    // ---- circuit output = level 1 (<= 2 rows per interaction): index 2 i + r, padding (0, 1)
    std::vector<Ext> out_n(2 * (size_t)W, kb::ext_zero()), out_d(2 * (size_t)W, kb::ext_one());
    {
        std::vector<Ext> hn(std::max<size_t>(level_entries[1], 1)), hd(std::max<size_t>(level_entries[1], 1));
        if (L >= 2) {
            SP1HIP_TRY(mb.fetch(lvN[1].p, level_entries[1] * 4, hn.data()));
        } else {                                             // L == 1: level 1 is the first layer itself (base numerators)
            std::vector<uint32_t> hb(std::max<size_t>(level_entries[1], 1));
            SP1HIP_TRY(mb.fetch(lvN[1].p, level_entries[1], hb.data()));
            for (size_t e = 0; e < level_entries[1]; e++) hn[e] = kb::ext_from_base(hb[e]);
        }
        SP1HIP_TRY(mb.fetch(lvD[1].p, level_entries[1] * 4, hd.data()));
        for (uint32_t i = 0; i < K; i++)
            for (uint32_t r = 0; r < rows_at(info[int_chip[i]].rows, 1); r++) { out_n[2 * i + r] = hn[off[1][i] + r]; out_d[2 * i + r] = hd[off[1][i] + r]; }
    }
    mark("circuit output fetched");
    challenger_observe(ch, kb::to_monty(2 * W));
    for (auto& e : out_n) observe_ext(ch, e);
    challenger_observe(ch, kb::to_monty(2 * W));
    for (auto& e : out_d) observe_ext(ch, e);
    std::vector<Ext> eval_point(niv + 1);
    for (auto& z : eval_point) z = challenger_sample_ext(ch);
    Ext num_eval = eval_mle_host(out_n, eval_point), den_eval = eval_mle_host(out_d, eval_point);

    size_t launch_idx = 0;                                   // next K descriptors of d_all
    RoundSyncHost rsync;
    SP1HIP_TRY(rsync.init(s));
    const Ext one = kb::ext_one(), inv8 = kb::ext_inv(ext_c(8)), inv2 = kb::ext_inv(ext_c(2)), four = ext_c(4);
    uint32_t h_sums[12];
    // device-resident transcript for the row rounds: SP1HIP_GKR_CHAIN=1. Byte-identical proofs (the GPU tests run both),
    // but OFF by default: measured on MI355X the round finished by one wave of the last workgroup (three cooperative
    // Poseidon2 permutations + the cubic: ~18 us of dependent instructions) costs what it saves — the host round trip
    // through mapped pinned memory is ~19 us per round (core-shaped shard: row rounds 32.3 ms chained vs 29.0 ms).
    const bool chain_enabled = [] { const char* e = getenv("SP1HIP_GKR_CHAIN"); return e && e[0] == '1'; }();   // read per call
    DeviceBuf d_chain, d_rconst, d_rout;
    std::vector<std::unique_ptr<std::vector<GkrRoundConst>>> keep_rconst;       // upload sources live to the end of the call
    std::vector<std::unique_ptr<GkrChain>> keep_chain;
    const p2::RoundConstants* d_p2rc = nullptr;
    if (chain_enabled) {
        const DeviceCtx* ctx;
        SP1HIP_TRY(get_device_ctx(&ctx));
        d_p2rc = ctx->d_rc;
        SP1HIP_TRY(d_chain.alloc(sizeof(GkrChain), s));
        SP1HIP_TRY(d_rconst.alloc(sizeof(GkrRoundConst) * (size_t)std::max(L, 1), s));
        SP1HIP_TRY(d_rout.alloc(sizeof(GkrRoundOut) * (size_t)std::max(L, 1), s));
    }

    HostPar::Scope par;                                      // helper threads for the host loops between hand-overs
    double dbg_rows = 0, dbg_int = 0, dbg_head = 0;
    auto dbg_t = std::chrono::steady_clock::now();
    for (int v = 1; v <= L - 1; v++) {
        if (gkr_debug) dbg_t = std::chrono::steady_clock::now();
        const Ext lambda = challenger_sample_ext(ch);
        RoundOut ro;
        Ext claim = num_eval * lambda + den_eval;
        ro.claimed_sum = claim;
        const std::vector<Ext> int_point(eval_point.begin(), eval_point.begin() + niv), row_point(eval_point.begin() + niv, eval_point.end());
        // Lagrange tables of every prefix of the interaction point (eq_tabs[m]: the first m coordinates, 2^m entries)
        std::vector<std::vector<Ext>> eq_tabs(niv + 1);
        eq_tabs[0] = {one};
        for (int m = 0; m < niv; m++) {
            const std::vector<Ext>& ev = eq_tabs[m];
            std::vector<Ext>& nx = eq_tabs[m + 1];
            nx.resize(ev.size() * 2);
            const Ext x = int_point[m];
            par.run(ev.size(), 64, [&](int, size_t b, size_t e) {
                for (size_t i = b; i < e; i++) { const Ext pr = ev[i] * x; nx[2 * i] = ev[i] - pr; nx[2 * i + 1] = pr; }
            });
        }
        const std::vector<Ext>& eq_int = eq_tabs[niv];
        par.park();                                          // the helpers sleep through the device rounds of the layer
        SP1HIP_TRY(stage.upload(d_eq_int.p, eq_int.data(), (size_t)W * 16));
        PointArg pa{};
        for (int j = 0; j < v; j++) pa.c[j] = row_point[j];
        hipLaunchKernelGGL(eq_prefix_tables_kernel, dim3(((2u << v) + 255) / 256), dim3(256), 0, s, pa, v, d_T.ext());
        SP1HIP_LAUNCH_CHECK();
        auto T_of = [&](int t) -> const Ext* { return d_T.ext() + (((size_t)1 << t) - 1); };

        std::vector<Ext> alphas;
        Ext PA = one;                                        // eq factor of the row variables bound so far
        Poly4 poly{};
        Ext alpha_r = kb::ext_zero();
        const bool chain = chain_enabled;
        if (chain) {
            // what the device needs to finish a round by itself: the sponge, the running claim, and per round the point
            // coordinate and the Lagrange basis of the nodes {0, 1, 1/2, (1 - pt) / (1 - 2 pt)} (they depend on the layer's
            // point only; the fourth node's value is zero, so three basis cubics suffice)
            keep_rconst.emplace_back(new std::vector<GkrRoundConst>(v));
            std::vector<GkrRoundConst>& rcs = *keep_rconst.back();
            for (int j = 0; j < v; j++) {
                const Ext pt = row_point[v - j - 1];
                const Ext xs[4] = {kb::ext_zero(), one, inv2, (one - pt) * kb::ext_inv(one - (pt + pt))};
                rcs[j].pt = pt;
                for (int k = 0; k < 3; k++) {
                    Ext ys[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
                    ys[k] = one;
                    const Poly4 b = interpolate4(xs, ys);
                    for (int d = 0; d < 4; d++) rcs[j].basis[k][d] = b[d];
                }
            }
            keep_chain.emplace_back(new GkrChain());
            GkrChain& hc = *keep_chain.back();
            uint32_t w34[34];
            challenger_export(ch, w34);
            memcpy(hc.ch_state, w34, 64); memcpy(hc.ch_in, w34 + 16, 32); memcpy(hc.ch_out, w34 + 25, 32);
            hc.ch_n_in = w34[24]; hc.ch_n_out = w34[33]; hc.pad0 = hc.pad1 = 0;
            hc.claim = claim; hc.PA = one; hc.alpha = kb::ext_zero();
            SP1HIP_TRY(stage.upload(d_chain.p, &hc, sizeof hc));
            SP1HIP_TRY(stage.upload(d_rconst.p, rcs.data(), sizeof(GkrRoundConst) * (size_t)v));
        }
        // per-interaction live row counts of the current round
        std::vector<uint32_t> live(K);
        uint32_t max_live = 0;
        for (uint32_t i = 0; i < K; i++) { live[i] = (rows_at(info[int_chip[i]].rows, v + 1) + 1) / 2; max_live = std::max(max_live, live[i]); }
        int cur = 0;
        std::vector<size_t> so_prev, so_next;
        for (int j = 0; j < v; j++) {                        // row-variable rounds
            const int t = v - j;                             // remaining row variables
            uint32_t tiles;
            const LaunchShape shape = shapes[launch_idx];
            const RoundDesc* d_descs = (const RoundDesc*)d_all.p + (launch_idx++) * K;
            const GkrChainArgs ca = chain ? GkrChainArgs{(GkrChain*)d_chain.p, (const GkrRoundConst*)d_rconst.p + j, (GkrRoundOut*)d_rout.p + j, d_p2rc}
                                          : GkrChainArgs{nullptr, nullptr, nullptr, nullptr};
            const FlatArgs fa{(const uint16_t*)d_flat.p + shape.flat_off, shape.total_pairs};
            if (j == 0) {
                tiles = shape.tiles;
                ScopedTimer tm("gkr_round_sum_first", s);
                const RoundSync rs = chain ? rsync.chained() : rsync.next();
#define SP1HIP_GKR_SUM_FIRST(NB, FL) hipLaunchKernelGGL((round_sum_first<NB, FL>), dim3(tiles), dim3(256), 0, s, d_descs, (const Ext*)d_eq_int.p, T_of(t), lambda, d_partials.u32(), rs, rsync.seq, K, shape.tile_size, ca, fa)
                if (v + 1 == L) { if (shape.flat) SP1HIP_GKR_SUM_FIRST(true, true); else SP1HIP_GKR_SUM_FIRST(true, false); }
                else { if (shape.flat) SP1HIP_GKR_SUM_FIRST(false, true); else SP1HIP_GKR_SUM_FIRST(false, false); }
#undef SP1HIP_GKR_SUM_FIRST
            } else {
                // fold round j-1 with alpha_r into scratch[cur], summing round j
                so_next.assign(K + 1, 0);
                uint32_t max_out = 0;
                for (uint32_t i = 0; i < K; i++) { const uint32_t o = (live[i] + 1) / 2; so_next[i + 1] = so_next[i] + o; max_out = std::max(max_out, o); }
                tiles = shape.tiles;
                ScopedTimer tm("gkr_round_fold_sum", s);
                const RoundSync rs = chain ? rsync.chained() : rsync.next();
#define SP1HIP_GKR_FOLD_SUM(F, NB, FL) hipLaunchKernelGGL((round_fold_sum<F, NB, true, FL>), dim3(tiles), dim3(256), 0, s, d_descs, (const Ext*)d_eq_int.p, T_of(t), lambda, alpha_r, d_partials.u32(), rs, rsync.seq, K, shape.tile_size, ca, fa)
                if (j == 1 && v + 1 == L) { if (shape.flat) SP1HIP_GKR_FOLD_SUM(true, true, true); else SP1HIP_GKR_FOLD_SUM(true, true, false); }
                else if (j == 1) { if (shape.flat) SP1HIP_GKR_FOLD_SUM(true, false, true); else SP1HIP_GKR_FOLD_SUM(true, false, false); }
                else { if (shape.flat) SP1HIP_GKR_FOLD_SUM(false, false, true); else SP1HIP_GKR_FOLD_SUM(false, false, false); }
#undef SP1HIP_GKR_FOLD_SUM
                for (uint32_t i = 0; i < K; i++) live[i] = (live[i] + 1) / 2;
                so_prev = so_next;
                cur ^= 1;
            }
            SP1HIP_LAUNCH_CHECK();
            (void)tiles;
            if (chain) continue;                             // the round finishes itself on the device; the next launch follows
            SP1HIP_TRY(rsync.wait(h_sums, 12));
            Ext S0, Sh, Seq;
            memcpy(&S0, h_sums, 16); memcpy(&Sh, h_sums + 4, 16); memcpy(&Seq, h_sums + 8, 16);
            const Ext pt = row_point[t - 1];
            const Ext corr = one - Seq;                      // eq mass of the padding entries (before the PA factor)
            const Ext p0 = PA * (S0 + corr * (one - pt));
            const Ext ph = PA * (Sh + corr * four) * inv8;
            const Ext xs[4] = {kb::ext_zero(), one, inv2, (one - pt) * kb::ext_inv(one - (pt + pt))};
            const Ext ys[4] = {p0, claim - p0, ph, kb::ext_zero()};
            poly = interpolate4(xs, ys);
            ro.polys.push_back(poly);
            for (auto& c : poly) observe_ext(ch, c);
            alpha_r = challenger_sample_ext(ch);
            alphas.push_back(alpha_r);
            claim = poly_eval(poly, alpha_r);
            PA = PA * (pt * alpha_r + (one - pt) * (one - alpha_r));
        }
        // bind the last row variable: one value per (interaction, table), dense over 2^niv on the host
        std::vector<Ext> tn0(W, kb::ext_zero()), td0(W, one), tn1(W, kb::ext_zero()), td1(W, one);
        par.wake();                                          // their wake-up hides behind the last fold and its hand-over
        {
            so_next.assign(K + 1, 0);
            for (uint32_t i = 0; i < K; i++) so_next[i + 1] = so_next[i] + (live[i] + 1) / 2;
            uint32_t max_out = 0;
            for (uint32_t i = 0; i < K; i++) max_out = std::max<uint32_t>(max_out, (live[i] + 1) / 2);
            const LaunchShape shape = shapes[launch_idx];
            const RoundDesc* d_descs = (const RoundDesc*)d_all.p + (launch_idx++) * K;
            const uint32_t tiles = shape.tiles;
            (void)max_out;
            const GkrChainArgs ca = chain ? GkrChainArgs{(GkrChain*)d_chain.p, nullptr, nullptr, nullptr} : GkrChainArgs{nullptr, nullptr, nullptr, nullptr};
            if (v == 1 && v + 1 == L) hipLaunchKernelGGL((round_fold_sum<true, true, false, false>), dim3(tiles), dim3(256), 0, s, d_descs, (const Ext*)d_eq_int.p, (const Ext*)nullptr, lambda, alpha_r, d_partials.u32(), RoundSync{}, 0u, K, shape.tile_size, ca, FlatArgs{nullptr, 0u});
            else if (v == 1) hipLaunchKernelGGL((round_fold_sum<true, false, false, false>), dim3(tiles), dim3(256), 0, s, d_descs, (const Ext*)d_eq_int.p, (const Ext*)nullptr, lambda, alpha_r, d_partials.u32(), RoundSync{}, 0u, K, shape.tile_size, ca, FlatArgs{nullptr, 0u});
            else hipLaunchKernelGGL((round_fold_sum<false, false, false, false>), dim3(tiles), dim3(256), 0, s, d_descs, (const Ext*)d_eq_int.p, (const Ext*)nullptr, lambda, alpha_r, d_partials.u32(), RoundSync{}, 0u, K, shape.tile_size, ca, FlatArgs{nullptr, 0u});
            SP1HIP_LAUNCH_CHECK();
            if (chain) {
                // ONE hand-over for the layer's v rounds: the messages, the challenges, the running values and the sponge
                std::vector<GkrRoundOut> outs(v);
                GkrChain hc;
                SP1HIP_TRY(mb.fetch(d_rout.p, sizeof(GkrRoundOut) / 4 * (size_t)v, outs.data()));
                SP1HIP_TRY(mb.fetch(d_chain.p, sizeof(GkrChain) / 4, &hc));
                rsync.settled();                                 // every chained launch of this layer has completed
                for (int j = 0; j < v; j++) {
                    Poly4 pj;
                    for (int d = 0; d < 4; d++) pj[d] = outs[j].poly[d];
                    ro.polys.push_back(pj);
                    alphas.push_back(outs[j].alpha);
                }
                uint32_t w34[34];
                memcpy(w34, hc.ch_state, 64); memcpy(w34 + 16, hc.ch_in, 32); w34[24] = hc.ch_n_in;
                memcpy(w34 + 25, hc.ch_out, 32); w34[33] = hc.ch_n_out;
                challenger_import(ch, w34);
                claim = hc.claim; PA = hc.PA; alpha_r = hc.alpha;
            }
            std::vector<Ext> host(std::max<size_t>(so_next[K], 1) * 4);
            SP1HIP_TRY(mb.fetch(scratch[cur].p, so_next[K] * 16, host.data()));
            for (uint32_t i = 0; i < K; i++) {
                if (so_next[i + 1] == so_next[i]) continue;   // chip without rows: stays (0, 1)
                const size_t base = 4 * so_next[i], len = so_next[i + 1] - so_next[i];
                tn0[i] = host[base]; td0[i] = host[base + len]; tn1[i] = host[base + 2 * len]; td1[i] = host[base + 3 * len];
            }
        }
        if (gkr_debug) { const auto now = std::chrono::steady_clock::now(); dbg_rows += std::chrono::duration<double, std::milli>(now - dbg_t).count(); dbg_t = now; }
        // interaction-variable rounds on the host (InteractionLayer, logup_poly.rs:L240-L316); eq_adjustment = PA.
        // The eq table is never folded: after binding the last j variables it is eq_scale x the Lagrange table of the
        // first niv - j coordinates (an intermediate table of the construction above), and a Lagrange table sums to one,
        // which gives the padding entries' share of both sums without visiting them. The sums and the folds of a round
        // run on the helper threads (host_par.hpp); the tables ping-pong because a parallel fold cannot be in place.
        Ext eq_scale = one;
        size_t real = K;                                     // entries >= real are the padding fraction (0, 1) in all four tables
        std::vector<Ext> un0(W / 2 + 1), ud0(W / 2 + 1), un1(W / 2 + 1), ud1(W / 2 + 1);
        std::vector<Ext>*tab[2][4] = {{&tn0, &td0, &tn1, &td1}, {&un0, &ud0, &un1, &ud1}};
        int side = 0;
        for (int j = 0; j < niv; j++) {
            const std::vector<Ext>& eqi = eq_tabs[niv - j];
            const size_t half = eqi.size() / 2;
            const size_t real_pairs = (real + 1) / 2;
            const Ext *n0 = tab[side][0]->data(), *d0 = tab[side][1]->data(), *n1 = tab[side][2]->data(), *d1 = tab[side][3]->data();
            struct alignas(64) Part { Ext x0, y0, xh, yh, e0, es; };
            Part parts[HostPar::Scope::MAX_THREADS];
            const int nparts_max = par.threads();
            for (int q = 0; q < nparts_max; q++) parts[q] = Part{kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
            par.run(real_pairs, 24, [&](int part, size_t kb0, size_t ke) {
                Part acc = parts[part];
                for (size_t k = kb0; k < ke; k++) {
                    const size_t a = 2 * k, b = 2 * k + 1;
                    // lambda is factored out of the sums: sum eq (lambda X + Y) = lambda sum eq X + sum eq Y
                    acc.x0 = acc.x0 + eqi[a] * (d0[a] * n1[a] + d1[a] * n0[a]);
                    acc.y0 = acc.y0 + eqi[a] * (d0[a] * d1[a]);
                    const Ext sn0 = n0[a] + n0[b], sn1 = n1[a] + n1[b], sd0 = d0[a] + d0[b], sd1 = d1[a] + d1[b];
                    const Ext es = eqi[a] + eqi[b];
                    acc.xh = acc.xh + es * (sd0 * sn1 + sd1 * sn0);
                    acc.yh = acc.yh + es * (sd0 * sd1);
                    acc.e0 = acc.e0 + eqi[a];
                    acc.es = acc.es + es;
                }
                parts[part] = acc;
            });
            Part t = parts[0];
            for (int q = 1; q < nparts_max; q++) {
                t.x0 = t.x0 + parts[q].x0; t.y0 = t.y0 + parts[q].y0; t.xh = t.xh + parts[q].xh; t.yh = t.yh + parts[q].yh;
                t.e0 = t.e0 + parts[q].e0; t.es = t.es + parts[q].es;
            }
            const Ext pt = int_point[niv - 1 - j];
            // a pair of padding entries contributes eq[a] * 1 to the first sum and (eq[a] + eq[b]) * (1 + 1)(1 + 1) to the
            // second; over ALL pairs sum eq[a] = 1 - pt and sum (eq[a] + eq[b]) = 1
            const Ext s0 = lambda * t.x0 + t.y0 + ((one - pt) - t.e0);
            const Ext sh = lambda * t.xh + t.yh + (one - t.es) * four;
            const Ext PAe = PA * eq_scale;
            const Ext p0 = PAe * s0, ph = PAe * sh * inv8;
            const Ext xs[4] = {kb::ext_zero(), one, inv2, (one - pt) * kb::ext_inv(one - (pt + pt))};
            const Ext ys[4] = {p0, claim - p0, ph, kb::ext_zero()};
            poly = interpolate4(xs, ys);
            ro.polys.push_back(poly);
            for (auto& c : poly) observe_ext(ch, c);
            alpha_r = challenger_sample_ext(ch);
            alphas.push_back(alpha_r);
            claim = poly_eval(poly, alpha_r);
            eq_scale = eq_scale * (pt * alpha_r + (one - pt) * (one - alpha_r));
            Ext *o0 = tab[side ^ 1][0]->data(), *o1 = tab[side ^ 1][1]->data(), *o2 = tab[side ^ 1][2]->data(), *o3 = tab[side ^ 1][3]->data();
            par.run(real_pairs, 48, [&](int, size_t kb0, size_t ke) {
                for (size_t k = kb0; k < ke; k++) {
                    o0[k] = n0[2 * k] + alpha_r * (n0[2 * k + 1] - n0[2 * k]); o1[k] = d0[2 * k] + alpha_r * (d0[2 * k + 1] - d0[2 * k]);
                    o2[k] = n1[2 * k] + alpha_r * (n1[2 * k + 1] - n1[2 * k]); o3[k] = d1[2 * k] + alpha_r * (d1[2 * k + 1] - d1[2 * k]);
                }
            });
            if (real_pairs < half) { o0[real_pairs] = kb::ext_zero(); o1[real_pairs] = one; o2[real_pairs] = kb::ext_zero(); o3[real_pairs] = one; }
            real = real_pairs;
            side ^= 1;
        }
        const Ext fin_n0 = (*tab[side][0])[0], fin_d0 = (*tab[side][1])[0], fin_n1 = (*tab[side][2])[0], fin_d1 = (*tab[side][3])[0];
        if (gkr_debug) { const auto now = std::chrono::steady_clock::now(); dbg_int += std::chrono::duration<double, std::milli>(now - dbg_t).count(); dbg_t = now; }
        ro.eval = claim;
        ro.point.assign(alphas.rbegin(), alphas.rend());
        ro.n0 = fin_n0; ro.d0 = fin_d0; ro.n1 = fin_n1; ro.d1 = fin_d1;
        observe_ext(ch, ro.n0); observe_ext(ch, ro.n1); observe_ext(ch, ro.d0); observe_ext(ch, ro.d1);
        eval_point = ro.point;
        const Ext lc = challenger_sample_ext(ch);
        num_eval = ro.n0 + (ro.n1 - ro.n0) * lc;
        den_eval = ro.d0 + (ro.d1 - ro.d0) * lc;
        eval_point.push_back(lc);
        rounds.push_back(std::move(ro));
    }

    if (gkr_debug) fprintf(stderr, "[sp1hip gkr]   of which row-variable rounds %.3f ms, interaction-variable rounds (host) %.3f ms\n", dbg_rows, dbg_int);
    (void)dbg_head;
    mark("all layers");
    // ---- trace openings at the last L coordinates
    const std::vector<Ext> trace_point(eval_point.end() - L, eval_point.end());
    std::vector<Ext> openings(std::max<size_t>(total_cols, 1), kb::ext_zero());
    if (total_cols) {
        DeviceBuf d_eq, d_od, d_part, d_res;
        SP1HIP_TRY(d_eq.alloc(((size_t)16) << L, s));
        SP1HIP_TRY(sp1hip_partial_lagrange(reinterpret_cast<const sp1hip_ext_t*>(trace_point.data()), L, d_eq.u32(), stream));
        std::vector<OpenDesc> od;
        uint32_t out0 = 0, max_rows_open = 0;
        for (int c = 0; c < n_chips; c++) {
            // proof order per chip: main evaluations, then preprocessed; opening slots: [main | prep]
            for (uint32_t g = 0; g < info[c].main_w; g += OPEN_COLS) od.push_back(OpenDesc{info[c].d_main, info[c].rows, info[c].main_w, g, out0 + g});
            out0 += info[c].main_w;
            for (uint32_t g = 0; g < info[c].prep_w; g += OPEN_COLS) od.push_back(OpenDesc{info[c].d_prep, info[c].rows, info[c].prep_w, g, out0 + g});
            out0 += info[c].prep_w;
            max_rows_open = std::max(max_rows_open, info[c].rows);
        }
        const uint32_t chunks = std::max<uint32_t>((max_rows_open + OPEN_ROWS - 1) / OPEN_ROWS, 1);
        SP1HIP_TRY(upload(d_od, od.data(), od.size() * sizeof(OpenDesc), s, stage));
        SP1HIP_TRY(d_part.alloc((size_t)chunks * total_cols * 16, s));
        SP1HIP_TRY(d_res.alloc(total_cols * 16, s));
        SP1HIP_HIP(hipMemsetAsync(d_part.p, 0, (size_t)chunks * total_cols * 16, s));
        ScopedTimer t("gkr_openings", s);
        hipLaunchKernelGGL(open_columns_kernel, dim3((uint32_t)od.size(), chunks), dim3(256), 0, s, (const OpenDesc*)d_od.p, d_eq.u32(),
                           1u << L, d_part.u32(), (uint32_t)total_cols);
        SP1HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(open_sum_kernel, dim3(((uint32_t)total_cols * 4 + 255) / 256), dim3(256), 0, s, d_part.u32(), chunks,
                           (uint32_t)total_cols * 4, d_res.u32());
        SP1HIP_LAUNCH_CHECK();
        SP1HIP_TRY(mb.fetch(d_res.p, total_cols * 4, openings.data()));
    }
    challenger_observe(ch, kb::to_monty((uint32_t)n_chips));
    {
        size_t o = 0;
        for (int c = 0; c < n_chips; c++) {
            const Ext* main = openings.data() + o;
            const Ext* prep = main + info[c].main_w;
            if (info[c].prep_w) {
                challenger_observe(ch, kb::to_monty(info[c].prep_w));
                for (uint32_t k = 0; k < info[c].prep_w; k++) observe_ext(ch, prep[k]);
            }
            challenger_observe(ch, kb::to_monty(info[c].main_w));
            for (uint32_t k = 0; k < info[c].main_w; k++) observe_ext(ch, main[k]);
            o += info[c].main_w + info[c].prep_w;
        }
    }

    mark("openings");
    // ---- bincode(LogupGkrProof)
    Bytes w;
    for (const std::vector<Ext>* vv : {&out_n, &out_d}) {
        w.u64(vv->size());
        for (auto& e : *vv) w.ext(e);
        w.u64(2); w.u64(vv->size()); w.u64(1);
    }
    w.u64(rounds.size());
    for (auto& r : rounds) {
        w.ext(r.n0); w.ext(r.n1); w.ext(r.d0); w.ext(r.d1);
        w.u64(r.polys.size());
        for (auto& p : r.polys) { w.u64(4); for (auto& c : p) w.ext(c); }
        w.ext(r.claimed_sum);
        w.u64(r.point.size());
        for (auto& x : r.point) w.ext(x);
        w.ext(r.eval);
    }
    w.u64(trace_point.size());
    for (auto& x : trace_point) w.ext(x);
    w.u64(n_chips);
    {
        size_t o = 0;
        for (int c = 0; c < n_chips; c++) {
            w.u64(info[c].name.size());
            for (char chr : info[c].name) w.b.push_back((uint8_t)chr);
            w.u64(info[c].main_w);
            for (uint32_t k = 0; k < info[c].main_w; k++) w.ext(openings[o + k]);
            w.u64(1); w.u64(info[c].main_w);
            w.b.push_back(info[c].prep_w ? 1 : 0);
            if (info[c].prep_w) {
                w.u64(info[c].prep_w);
                for (uint32_t k = 0; k < info[c].prep_w; k++) w.ext(openings[o + info[c].main_w + k]);
                w.u64(1); w.u64(info[c].prep_w);
            }
            o += info[c].main_w + info[c].prep_w;
        }
    }
    w.felt(witness);
    if (w.b.size() != need) {
        set_error("internal error: GKR proof size %zu != expected %zu", w.b.size(), need);
        return SP1HIP_ERROR_RUNTIME;
    }
    memcpy(h_proof, w.b.data(), w.b.size());
    *proof_len = w.b.size();
    challenger_restore(challenger, ch);
    return SP1HIP_SUCCESS;
}

}  // extern "C"
