// sp1_amd/csrc/jagged.hip — the jagged PCS evaluation proof: everything between the
// zerocheck point and the BaseFold opening, on the device.
//
//   sp1hip_jagged_prove   `JaggedProver::prove_trusted_evaluations`  /root/reference/slop/crates/jagged/src/prover.rs:L162-L328
//     jagged sumcheck     `jagged_sumcheck_poly` + `HadamardProduct`  sumcheck.rs:L13-L39, hadamard.rs:L52-L146
//                         driven as `reduce_sumcheck_to_evaluation`   /root/reference/slop/crates/sumcheck/src/prover.rs:L13-L96
//     J table             `partial_jagged_little_polynomial_evaluation` poly.rs:L258-L318
//     jagged-eval proof   `JaggedEvalSumcheckProver::prove_jagged_evaluation` jagged_eval/sumcheck_eval.rs:L185-L243,
//                         sumcheck_poly.rs, sumcheck_sum_as_poly.rs, eval_sumcheck_prover.rs; branching program poly.rs:L120-L160,L385-L460
//     dense opening       `StackedPcsProver::prove_trusted_evaluation` /root/reference/slop/crates/stacked/src/prover.rs:L107-L152
//                         + `prove_untrusted_evaluation(s)` (multilinear/src/pcs.rs:L128-L139, basefold-prover/src/prover.rs:L245-L270)
//   Output: bincode(JaggedPcsProof) (/root/reference/slop/crates/jagged/src/verifier.rs:L17-L26).
//
// MI355X shape
//  * The dense vector q of the sumcheck is the commit-time dense buffer itself (column-major stacked
//    batches ARE the long vector the reference re-stacks), read in place, one segment per round.
//  * J(x) = eq(z_col, col(x)) eq(z_row, row(x)) is never materialised at full length: round 0 and the
//    first fold recompute it from the two small eq tables (row table: 2^max_log_row_count ext, SoA,
//    coalesced along a column; column table + prefix sums: L2 resident; column of x by binary search).
//  * One fused kernel per round: fold the previous round's tables with alpha AND accumulate the next
//    round's y(0), 4 y(1/2) sums from the folded values while they are in registers (read 32 B, write
//    16 B per surviving element). Everything past the real area T is zero and is neither stored nor read.
//  * The jagged-eval sumcheck does NOT re-run the branching program per column, round and evaluation
//    point (the reference's CPU path: ~2(log m + 1) layers x 31 ext products each time). The program
//    is a product of 4x4 transfer matrices, one per bit layer, multilinear in the layer's two prefix
//    bits: per column we keep a running row vector (layers already bound to challenges) and
//    precomputed suffix vectors (layers still boolean), so a round costs two 4x4 vector-matrix
//    products and two dot products per column; in the second half of the rounds the bound part is
//    column-independent and is advanced on the host. Same field elements, ~30x less work.
#include <cstring>
#include <memory>
#include <vector>

#include <algorithm>
#include <array>

#include "device_ctx.hpp"
#include "round_sync.hpp"
#include "stacked_data.hpp"
#include "tensor_table.hpp"

namespace sp1hip {

struct DeviceBuf : AsyncScratch {
    uint32_t* u32() const { return (uint32_t*)p; }
};

void challenger_observe(sp1hip_challenger_t* ch, uint32_t x);
kb::Ext challenger_sample_ext(sp1hip_challenger_t* ch);
void challenger_restore(sp1hip_challenger_t* dst, const sp1hip_challenger_t* src);

using Ext = kb::Ext;

struct JgSeg { const uint32_t* ptr; uint32_t start, len; };
struct JgSegs { JgSeg s[8]; int n; uint32_t total; };

struct JgJ {                       // what J(x) needs
    const uint32_t* prefix;        // [ncols + 1]
    uint32_t ncols;
    const Ext* col_eq;             // [>= ncols] AoS
    const uint32_t* row_eq;        // SoA, 4 planes of row_len
    uint32_t row_len;
};

__device__ __forceinline__ Ext ld_ext(const Ext* p, uint32_t i) {
    const uint4 v = reinterpret_cast<const uint4*>(p)[i];
    return Ext{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void st_ext(Ext* p, uint32_t i, const Ext& e) {
    reinterpret_cast<uint4*>(p)[i] = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

__device__ __forceinline__ uint32_t jg_q(const JgSegs& S, uint32_t x) {
#pragma unroll 1
    for (int k = 0; k < S.n; k++)
        if (x - S.s[k].start < S.s[k].len) return S.s[k].ptr[x - S.s[k].start];
    return 0u;
}

// last column c with prefix[c] <= x  (then prefix[c + 1] > x because x < prefix[ncols])
__device__ __forceinline__ uint32_t jg_find_col(const JgJ& J, uint32_t x) {
    uint32_t lo = 0, hi = J.ncols;   // invariant: prefix[lo] <= x < prefix[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (J.prefix[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ Ext jg_j_at(const JgJ& J, uint32_t c, uint32_t row) {
    const Ext r{{J.row_eq[row], J.row_eq[J.row_len + row], J.row_eq[2 * J.row_len + row], J.row_eq[3 * J.row_len + row]}};
    return kb::ext_mul(ld_ext(J.col_eq, c), r);
}

// q and J for NV consecutive dense indices x0 .. x0 + NV - 1 (zero past `total`)
template <int NV>
__device__ __forceinline__ void jg_load_base(const JgSegs& S, const JgJ& J, uint32_t x0, uint32_t (&q)[NV], Ext (&j)[NV]) {
    uint32_t c = 0;
    bool have = false;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const uint32_t x = x0 + v;
        if (x >= S.total) { q[v] = 0u; j[v] = kb::ext_zero(); continue; }
        q[v] = jg_q(S, x);
        if (!have) { c = jg_find_col(J, x); have = true; }
        else while (J.prefix[c + 1] <= x) c++;
        j[v] = jg_j_at(J, c, x - J.prefix[c]);
    }
}

// ---- block reduction of two ext accumulators -> partials[block][8]
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ void block_reduce_store(const Ext& a, const Ext& b, uint32_t* out8) {
    __shared__ uint32_t sm[16][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { w[k] = wave_sum(a.c[k]); w[4 + k] = wave_sum(b.c[k]); }
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 8; k++) sm[wave][k] = w[k];
    __syncthreads();
    if (threadIdx.x < 8) {
        uint32_t acc = 0;
        for (int i = 0; i < nw; i++) acc = kb::add(acc, sm[i][threadIdx.x]);
        out8[threadIdx.x] = acc;
    }
}

// The end of every round kernel of this file: the two extension sums of the round go through the last-workgroup hand-over
// of round_sync.hpp (rs_finish: coherent partial stores, ticket, the last workgroup totals and publishes to mapped host
// memory) — the one-workgroup reduce kernel that used to follow each of the ~120 round launches of a proof is gone.
struct JgTail { uint32_t* partials; RoundSync rs; uint32_t seq; };
__device__ __forceinline__ void jg_finish(const Ext& a, const Ext& b, const JgTail& t) {
    const Ext acc[2] = {a, b};
    rs_finish<2>(acc, t.partials, blockIdx.x, gridDim.x, t.rs, t.seq);
}
__device__ __forceinline__ void jg_finish8(const Ext (&acc)[8], const JgTail& t) {      // the two-round pass: eight sums
    rs_finish<8>(acc, t.partials, blockIdx.x, gridDim.x, t.rs, t.seq);
}

// Sums the block partials and hands the two ext sums to the host through the mailbox slot (round_sync.hpp): payload
// words [1..8], then the sequence number — no copy, no stream synchronise.
__global__ __launch_bounds__(256) void jg_reduce_partials(const uint32_t* __restrict__ partials, uint32_t n,
                                                          volatile uint32_t* slot, uint32_t seq) {
    __shared__ uint32_t out8[8];
    Ext a = kb::ext_zero(), b = kb::ext_zero();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        a = kb::ext_add(a, Ext{{partials[8 * i], partials[8 * i + 1], partials[8 * i + 2], partials[8 * i + 3]}});
        b = kb::ext_add(b, Ext{{partials[8 * i + 4], partials[8 * i + 5], partials[8 * i + 6], partials[8 * i + 7]}});
    }
    block_reduce_store(a, b, out8);
    if (threadIdx.x < 8) slot[1 + threadIdx.x] = out8[threadIdx.x];     // the same lanes wrote out8
    // the slot is uncached host memory: acknowledged stores are ordered before the sequence number; a system-scope fence
    // would also write back this XCD's whole L2 (round_sync.hpp)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) slot[0] = seq;
}

// ---- round 0: y(0) = sum J[2i] q[2i], H = sum (J[2i] + J[2i+1]) (q[2i] + q[2i+1])   (hadamard.rs:L100-L135)
// J(x) = eq_col[c] * eq_row[row] and a column is a long run of consecutive x, so eq_col[c] is factored out of
// the run: per element only ext x base products (4 multiplies) remain, and the column search is done once per
// thread. A thread owns RUN consecutive dense indices; when its run crosses into another column the partial
// sums are flushed through that column's eq_col. A pair that straddles two columns takes the unfactored formula.
constexpr int JG_ITERS = 16;     // pairs per thread; a workgroup covers 256 * JG_ITERS consecutive pairs
__device__ __forceinline__ Ext jg_row_eq(const JgJ& J, uint32_t row) {
    return Ext{{J.row_eq[row], J.row_eq[J.row_len + row], J.row_eq[2 * J.row_len + row], J.row_eq[3 * J.row_len + row]}};
}
// Lanes take CONSECUTIVE pairs (coalesced 8 B of q and 2 x 4 planes of eq_row per lane) and step by 256 pairs,
// so a lane stays inside one column for many steps and keeps that column's sums un-multiplied.
__global__ __launch_bounds__(256) void jg_round0_sum(JgSegs S, JgJ J, uint32_t n_pairs, JgTail tail) {
    Ext e0 = kb::ext_zero(), eh = kb::ext_zero();
    for (uint32_t chunk = blockIdx.x; (uint64_t)chunk * 256 * JG_ITERS < n_pairs; chunk += gridDim.x) {
        uint32_t c = 0, col_end = 0;
        bool have = false;
        Ext a0 = kb::ext_zero(), ah = kb::ext_zero();     // sums of the current column, without eq_col[c]
#pragma unroll 1
        for (int it = 0; it < JG_ITERS; it++) {
            const uint32_t pair = chunk * 256 * JG_ITERS + it * 256 + threadIdx.x;
            if (pair >= n_pairs) break;
            const uint32_t x = 2 * pair;
            if (!have) { c = jg_find_col(J, x); col_end = J.prefix[c + 1]; have = true; }
            else if (x >= col_end) {                      // entered a later column: flush
                const Ext w = ld_ext(J.col_eq, c);
                e0 = kb::ext_add(e0, kb::ext_mul(w, a0));
                eh = kb::ext_add(eh, kb::ext_mul(w, ah));
                a0 = ah = kb::ext_zero();
                while (J.prefix[c + 1] <= x) c++;
                col_end = J.prefix[c + 1];
            }
            const uint32_t q0 = jg_q(S, x), q1 = jg_q(S, x + 1);
            const uint32_t row = x - J.prefix[c];
            const Ext r0 = jg_row_eq(J, row);
            a0 = kb::ext_add(a0, kb::ext_mul_base(r0, q0));
            if (x + 1 < col_end) {
                ah = kb::ext_add(ah, kb::ext_mul_base(kb::ext_add(r0, jg_row_eq(J, row + 1)), kb::add(q0, q1)));
            } else {                                      // x + 1 opens the next non-empty column (at its row 0)
                uint32_t c1 = c + 1;
                while (J.prefix[c1 + 1] <= x + 1) c1++;
                const Ext j0 = kb::ext_mul(ld_ext(J.col_eq, c), r0), j1 = kb::ext_mul(ld_ext(J.col_eq, c1), jg_row_eq(J, x + 1 - J.prefix[c1]));
                eh = kb::ext_add(eh, kb::ext_mul_base(kb::ext_add(j0, j1), kb::add(q0, q1)));
            }
        }
        if (have) {
            const Ext w = ld_ext(J.col_eq, c);
            e0 = kb::ext_add(e0, kb::ext_mul(w, a0));
            eh = kb::ext_add(eh, kb::ext_mul(w, ah));
        }
    }
    jg_finish(e0, eh, tail);
}

// ---- round 0, table-major form (every table height even, so dense pairs are row pairs (2k, 2k+1) of one column).
// sum_x J(x) q(x) = sum_tables sum_rows eq_row[r] * (sum_c eq_col[c] q[c, r]): a lane owns one row pair of a table
// and walks the table's columns — consecutive lanes read consecutive 8-byte pairs of a column — accumulating
// sum_c eq_col[c] q unreduced (kb::DotAcc, wave-uniform coefficients), and multiplies by eq_row ONCE per row pair.
// The generic kernel above re-reads the 16 B/row eq_row table for every column (6.4 GB at core scale against the
// 1.6 GB of q) and spends an ext x base product plus a table walk per element.
struct JgTab { const uint32_t* q; uint32_t height, col0, ncols, tile0, x0, tile1, tile2; };   // q: the table's first column in the dense buffer; x0: its dense index; tile0/1/2: first tile in the round-0 / one-level / two-level fold launches
constexpr uint32_t JG_TAB_PAIRS = 256;                                       // row pairs per tile
__global__ __launch_bounds__(256) void jg_round0_tables(const JgTab* __restrict__ tabs, uint32_t n_tabs, uint32_t n_tiles,
                                                        JgJ J, JgTail tail) {
    Ext e0 = kb::ext_zero(), eh = kb::ext_zero();
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t lo = 0, hi = n_tabs;                     // last table with tile0 <= tile
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (__builtin_amdgcn_readfirstlane(tabs[mid].tile0) <= tile) lo = mid; else hi = mid;
        }
        const JgTab t = tabs[lo];
        const uint32_t k = (tile - t.tile0) * JG_TAB_PAIRS + threadIdx.x;     // row pair of this lane
        if (2 * k >= t.height) continue;
        kb::DotAcc u0, us;
        kb::dot_init(u0);
        kb::dot_init(us);
        const uint32_t* col = t.q + 2 * (size_t)k;
        for (uint32_t c = 0; c < t.ncols; c++, col += t.height) {
            if ((c & 0x3fffu) == 0x3fffu) {               // 2^14 columns per accumulator window (never in practice)
                const Ext r0 = jg_row_eq(J, 2 * k), r1 = jg_row_eq(J, 2 * k + 1);
                e0 = kb::ext_add(e0, kb::ext_mul(r0, kb::dot_finish(u0)));
                eh = kb::ext_add(eh, kb::ext_mul(kb::ext_add(r0, r1), kb::dot_finish(us)));
                kb::dot_init(u0);
                kb::dot_init(us);
            }
            const uint2 v = *reinterpret_cast<const uint2*>(col);
            const Ext w = ld_ext(J.col_eq, t.col0 + c);   // wave-uniform
            kb::dot_add(u0, w, v.x);
            kb::dot_add(us, w, kb::add(v.x, v.y));
        }
        const Ext r0 = jg_row_eq(J, 2 * k), r1 = jg_row_eq(J, 2 * k + 1);
        e0 = kb::ext_add(e0, kb::ext_mul(r0, kb::dot_finish(u0)));
        eh = kb::ext_add(eh, kb::ext_mul(kb::ext_add(r0, r1), kb::dot_finish(us)));
    }
    jg_finish(e0, eh, tail);
}

// ---- rounds 0 AND 1 from one pass over the base words (table heights % 4 == 0; the look-ahead of the reference's
// /root/reference/sp1-gpu/crates/sys/lib/experimental/look_ahead.cu:L98 on this file's table-major form). A lane owns base
// rows 4k .. 4k+3 of a table — the values of q and J at (X, Y) = (bit 0, bit 1) in {0,1}^2 — and walks the columns once,
// accumulating U_l = sum_c eq_col[c] q[c, 4k + l] unreduced like round 0 does (the same four multiply-adds per element as
// the two accumulators of jg_round0_tables on a row pair). J(x) = eq_col[c] r_l with r_l = eq_row[4k + l], and everything
// the two rounds need is bilinear in (r, U):
//   round 0:  y(0) = sum r_0 U_0 + r_2 U_2,   H = sum (r_0 + r_1)(U_0 + U_1) + (r_2 + r_3)(U_2 + U_3)
//   round 1 (after alpha_0, with r'(Y) = r_0Y + alpha_0 (r_1Y - r_0Y) and U' likewise):
//             y'(0) = sum r'(0) U'(0):  a quadratic in alpha_0 through P0 = r_0 U_0, P1 = r_1 U_1, leading coefficient
//                     (r_1 - r_0)(U_1 - U_0) = 2 (P0 + P1) - (r_0 + r_1)(U_0 + U_1)
//             H'    = sum (r'(0) + r'(1))(U'(0) + U'(1)):  through Q0 = (r_0 + r_2)(U_0 + U_2), Q1 = (r_1 + r_3)(U_1 + U_3),
//                     leading coefficient Qinf = ((r_1 + r_3) - (r_0 + r_2))((U_1 + U_3) - (U_0 + U_2))
// Eight sums: [P0, r_2 U_2, Ha, Hb, P1, Q0, Q1, Qinf] — eight extension products per lane and table, not per element. The
// second pass over the 1.6 GB of base words that round 1 used to be (jg_fold_tables<1, false>, sums only) is gone.
__global__ __launch_bounds__(256) void jg_round01_tables(const JgTab* __restrict__ tabs, uint32_t n_tabs, uint32_t n_tiles,
                                                         JgJ J, JgTail tail) {
    Ext acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = kb::ext_zero();
    auto flush = [&](const kb::DotAcc (&u)[4], uint32_t k) {
        Ext U[4], r[4];
#pragma unroll
        for (int l = 0; l < 4; l++) { U[l] = kb::dot_finish(u[l]); r[l] = jg_row_eq(J, 4 * k + l); }
        acc[0] = kb::ext_add(acc[0], kb::ext_mul(r[0], U[0]));
        acc[1] = kb::ext_add(acc[1], kb::ext_mul(r[2], U[2]));
        acc[2] = kb::ext_add(acc[2], kb::ext_mul(kb::ext_add(r[0], r[1]), kb::ext_add(U[0], U[1])));
        acc[3] = kb::ext_add(acc[3], kb::ext_mul(kb::ext_add(r[2], r[3]), kb::ext_add(U[2], U[3])));
        acc[4] = kb::ext_add(acc[4], kb::ext_mul(r[1], U[1]));
        const Ext sr0 = kb::ext_add(r[0], r[2]), sr1 = kb::ext_add(r[1], r[3]), sU0 = kb::ext_add(U[0], U[2]), sU1 = kb::ext_add(U[1], U[3]);
        acc[5] = kb::ext_add(acc[5], kb::ext_mul(sr0, sU0));
        acc[6] = kb::ext_add(acc[6], kb::ext_mul(sr1, sU1));
        acc[7] = kb::ext_add(acc[7], kb::ext_mul(kb::ext_sub(sr1, sr0), kb::ext_sub(sU1, sU0)));
    };
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t lo = 0, hi = n_tabs;                     // last table with tile1 <= tile
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (__builtin_amdgcn_readfirstlane(tabs[mid].tile1) <= tile) lo = mid; else hi = mid;
        }
        const JgTab t = tabs[lo];
        const uint32_t k = (tile - t.tile1) * JG_TAB_PAIRS + threadIdx.x;     // row quad of this lane
        if (4 * k >= t.height) continue;
        kb::DotAcc u[4];
#pragma unroll
        for (int l = 0; l < 4; l++) kb::dot_init(u[l]);
        const uint32_t* col = t.q + 4 * (size_t)k;
        for (uint32_t c = 0; c < t.ncols; c++, col += t.height) {
            if ((c & 0x3fffu) == 0x3fffu) {               // 2^14 columns per accumulator window (never in practice)
                flush(u, k);
#pragma unroll
                for (int l = 0; l < 4; l++) kb::dot_init(u[l]);
            }
            const uint4 v = *reinterpret_cast<const uint4*>(col);
            const Ext w = ld_ext(J.col_eq, t.col0 + c);   // wave-uniform
            kb::dot_add(u[0], w, v.x);
            kb::dot_add(u[1], w, v.y);
            kb::dot_add(u[2], w, v.z);
            kb::dot_add(u[3], w, v.w);
        }
        flush(u, k);
    }
    jg_finish8(acc, tail);
}

__device__ __forceinline__ Ext fold_ext(const Ext& a, const Ext& b, const Ext& alpha) {   // a + alpha (b - a)
    return kb::ext_add(a, kb::ext_mul(kb::ext_sub(b, a), alpha));      // alpha: wave-uniform, second
}

// ---- first fold(s), table-major form (J stays factored). LEVELS = 1 (heights % 4 == 0): a lane owns the level-1 row
// pair (2k, 2k+1) of a table = base rows 4k .. 4k+3 of every column; q1 = lerp of the base pairs with alpha0.
// LEVELS = 2 (heights % 8 == 0): the level-2 row pair = base rows 8k .. 8k+7; level 1 is recomputed in registers and
// folded again with alpha1, so the level-1 table (16 B per entry written, then read) never exists: round 1 runs
// with STORE = false (sums only, 4 B/element read), round 2 reads the base words again and writes level 2.
// The next round's sums use sum_c eq_col[c] q[c, r] accumulated unreduced (kb::edot_add with the (eq_col, 3 eq_col)
// pairs), times the level's eq_row once per row.
template <int LEVELS, bool STORE>
__global__ __launch_bounds__(256) void jg_fold_tables(const JgTab* __restrict__ tabs, uint32_t n_tabs, uint32_t n_tiles, JgJ Jl,
                                                      const Ext* __restrict__ col_eq3, Ext alpha0, Ext alpha1, Ext* __restrict__ q_out,
                                                      JgTail tail) {
    constexpr uint32_t SPAN = 2u << LEVELS;               // base rows per lane and column
    Ext e0 = kb::ext_zero(), eh = kb::ext_zero();
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t lo = 0, hi = n_tabs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint32_t t0 = __builtin_amdgcn_readfirstlane(LEVELS == 1 ? tabs[mid].tile1 : tabs[mid].tile2);
            if (t0 <= tile) lo = mid; else hi = mid;
        }
        const JgTab t = tabs[lo];
        const uint32_t k = (tile - (LEVELS == 1 ? t.tile1 : t.tile2)) * JG_TAB_PAIRS + threadIdx.x;   // row pair of this lane
        if (SPAN * k >= t.height) continue;
        kb::DotAcc u0, us;
        kb::dot_init(u0);
        kb::dot_init(us);
        const uint32_t* col = t.q + SPAN * (size_t)k;
        const uint32_t hl = t.height >> LEVELS;
        Ext* out = q_out + (t.x0 >> LEVELS) + 2 * (size_t)k;
        for (uint32_t c = 0; c < t.ncols; c++, col += t.height, out += hl) {
            if ((c & 0xfffu) == 0xfffu) {                 // 2^12 columns per accumulator window (never in practice)
                const Ext r0 = jg_row_eq(Jl, 2 * k), r1 = jg_row_eq(Jl, 2 * k + 1);
                e0 = kb::ext_add(e0, kb::ext_mul(r0, kb::dot_finish(u0)));
                eh = kb::ext_add(eh, kb::ext_mul(kb::ext_add(r0, r1), kb::dot_finish(us)));
                kb::dot_init(u0);
                kb::dot_init(us);
            }
            Ext qa, qb;
            const uint4 v = *reinterpret_cast<const uint4*>(col);
            const Ext l0 = kb::ext_add(kb::ext_from_base(v.x), kb::ext_mul_base(alpha0, kb::sub(v.y, v.x)));
            const Ext l1 = kb::ext_add(kb::ext_from_base(v.z), kb::ext_mul_base(alpha0, kb::sub(v.w, v.z)));
            if (LEVELS == 1) { qa = l0; qb = l1; }
            else {
                const uint4 v2 = *reinterpret_cast<const uint4*>(col + 4);
                const Ext l2 = kb::ext_add(kb::ext_from_base(v2.x), kb::ext_mul_base(alpha0, kb::sub(v2.y, v2.x)));
                const Ext l3 = kb::ext_add(kb::ext_from_base(v2.z), kb::ext_mul_base(alpha0, kb::sub(v2.w, v2.z)));
                qa = kb::ext_add(l0, kb::ext_mul(kb::ext_sub(l1, l0), alpha1));
                qb = kb::ext_add(l2, kb::ext_mul(kb::ext_sub(l3, l2), alpha1));
            }
            if (STORE) { st_ext(out, 0, qa); st_ext(out, 1, qb); }
            const Ext w = ld_ext(Jl.col_eq, t.col0 + c), w3 = ld_ext(col_eq3, t.col0 + c);     // wave-uniform
            kb::edot_add(u0, w, w3, qa);
            kb::edot_add(us, w, w3, kb::ext_add(qa, qb));
        }
        const Ext r0 = jg_row_eq(Jl, 2 * k), r1 = jg_row_eq(Jl, 2 * k + 1);
        e0 = kb::ext_add(e0, kb::ext_mul(r0, kb::dot_finish(u0)));
        eh = kb::ext_add(eh, kb::ext_mul(kb::ext_add(r0, r1), kb::dot_finish(us)));
    }
    jg_finish(e0, eh, tail);
}

// ---- first fold (base q, recomputed J) fused with the next round's sums. One step of a thread: inputs 4k..4k+3,
// outputs 2k, 2k+1 of the round-1 tables (n_out entries). Lanes take consecutive k (16 B of q per lane) and step by
// 256, tracking their column like jg_round0_sum; inside one column J folds as eq_col[c] * lerp(eq_row[r], eq_row[r+1]).
// WRITE_J = false when the next level keeps J factored (jg_foldf_sum): the 16 B/entry j table is never materialised.
template <bool WRITE_J>
__global__ __launch_bounds__(256) void jg_fold0_sum(JgSegs S, JgJ J, Ext alpha, uint32_t n_out, Ext* __restrict__ q_out,
                                                    Ext* __restrict__ j_out, JgTail tail) {
    Ext e0 = kb::ext_zero(), eh = kb::ext_zero();
    const uint32_t n_thr = (n_out + 1) / 2;
    for (uint32_t chunk = blockIdx.x; (uint64_t)chunk * 256 * JG_ITERS < n_thr; chunk += gridDim.x) {
        uint32_t c = 0, col_end = 0;
        bool have = false;
#pragma unroll 1
        for (int it = 0; it < JG_ITERS; it++) {
            const uint32_t k = chunk * 256 * JG_ITERS + it * 256 + threadIdx.x;
            if (k >= n_thr) break;
            const uint32_t x0 = 4 * k;
            Ext qo[2], jo[2];
            if (x0 + 3 < S.total) {
                if (!have) { c = jg_find_col(J, x0); col_end = J.prefix[c + 1]; have = true; }
                else if (x0 >= col_end) { while (J.prefix[c + 1] <= x0) c++; col_end = J.prefix[c + 1]; }
            }
            if (x0 + 3 < S.total && x0 + 3 < col_end) {       // all four inputs in column c
                const uint32_t row = x0 - J.prefix[c];
                const Ext w = ld_ext(J.col_eq, c);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t qa = jg_q(S, x0 + 2 * h), qb = jg_q(S, x0 + 2 * h + 1);
                    qo[h] = kb::ext_add(kb::ext_from_base(qa), kb::ext_mul_base(alpha, kb::sub(qb, qa)));
                    const Ext ra = jg_row_eq(J, row + 2 * h), rb = jg_row_eq(J, row + 2 * h + 1);
                    jo[h] = kb::ext_mul(w, kb::ext_add(ra, kb::ext_mul(kb::ext_sub(rb, ra), alpha)));
                }
            } else {                                          // column boundary or the zero tail: element by element
                uint32_t q[4];
                Ext j[4];
                jg_load_base<4>(S, J, x0, q, j);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    qo[h] = kb::ext_add(kb::ext_from_base(q[2 * h]), kb::ext_mul_base(alpha, kb::sub(q[2 * h + 1], q[2 * h])));
                    jo[h] = fold_ext(j[2 * h], j[2 * h + 1], alpha);
                }
                have = false;                                 // re-search next time (cheap: boundaries are rare)
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (2 * k + h < n_out) { st_ext(q_out, 2 * k + h, qo[h]); if (WRITE_J) st_ext(j_out, 2 * k + h, jo[h]); }
            e0 = kb::ext_add(e0, kb::ext_mul(jo[0], qo[0]));
            eh = kb::ext_add(eh, kb::ext_mul(kb::ext_add(jo[0], jo[1]), kb::ext_add(qo[0], qo[1])));
        }
    }
    jg_finish(e0, eh, tail);
}

// ---- folds that keep J factored. While every column starts at a multiple of 2^r in the dense order (chip heights
// are multiples of 32 in the reference's shards, powers of two in the synthetic ones), folding the r lowest dense
// variables never mixes two columns, so J_r(x) = eq_col[c] * eq_row_r[x - (prefix[c] >> r)] with eq_row_r the row
// table folded r times (a 2^(L-r)-entry table that lives in L2 / MALL). The j tables of these levels — 32 B/entry
// written and read back, 12 GB over levels 1..5 at core scale — are never materialised: a fold reads q_{r-1}, writes
// q_r and sums against the factored J_r, with eq_col[c] pulled out of each column run as in round 0.
__global__ __launch_bounds__(256) void jg_fold_row_eq(const uint32_t* __restrict__ in, uint32_t len_in, Ext alpha,
                                                      uint32_t* __restrict__ out) {
    const uint32_t len_out = len_in >> 1;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= len_out) return;
    const Ext a{{in[2 * i], in[len_in + 2 * i], in[2 * len_in + 2 * i], in[3 * len_in + 2 * i]}};
    const Ext b{{in[2 * i + 1], in[len_in + 2 * i + 1], in[2 * len_in + 2 * i + 1], in[3 * len_in + 2 * i + 1]}};
    const Ext r = fold_ext(a, b, alpha);
#pragma unroll
    for (int k = 0; k < 4; k++) out[(size_t)k * len_out + i] = r.c[k];
}

// q_in: level r-1 (n_in live entries); J: level r (prefix >> r, eq_row_r); writes q_out = level r (n_out entries)
__global__ __launch_bounds__(256) void jg_foldf_sum(const Ext* __restrict__ q_in, uint32_t n_in, JgJ J, Ext alpha, uint32_t n_out,
                                                    Ext* __restrict__ q_out, JgTail tail) {
    Ext e0 = kb::ext_zero(), eh = kb::ext_zero();
    const uint32_t n_pairs = (n_out + 1) / 2;
    for (uint32_t chunk = blockIdx.x; (uint64_t)chunk * 256 * JG_ITERS < n_pairs; chunk += gridDim.x) {
        uint32_t c = 0, col_end = 0;
        bool have = false;
        Ext a0 = kb::ext_zero(), ah = kb::ext_zero();     // sums of the current column, without eq_col[c]
#pragma unroll 1
        for (int it = 0; it < JG_ITERS; it++) {
            const uint32_t pair = chunk * 256 * JG_ITERS + it * 256 + threadIdx.x;
            if (pair >= n_pairs) break;
            const uint32_t x = 2 * pair;                  // level-r index of the first output
            Ext q[4];
#pragma unroll
            for (int v = 0; v < 4; v++) q[v] = 2 * x + v < n_in ? ld_ext(q_in, 2 * x + v) : kb::ext_zero();
            const Ext q0 = fold_ext(q[0], q[1], alpha), q1 = fold_ext(q[2], q[3], alpha);
            st_ext(q_out, x, q0);
            if (x + 1 < n_out) st_ext(q_out, x + 1, q1);
            if (!have) { c = jg_find_col(J, x); col_end = J.prefix[c + 1]; have = true; }
            else if (x >= col_end) {                      // entered a later column: flush
                const Ext w = ld_ext(J.col_eq, c);
                e0 = kb::ext_add(e0, kb::ext_mul(w, a0));
                eh = kb::ext_add(eh, kb::ext_mul(w, ah));
                a0 = ah = kb::ext_zero();
                while (J.prefix[c + 1] <= x) c++;
                col_end = J.prefix[c + 1];
            }
            const Ext r0 = jg_row_eq(J, x - J.prefix[c]);
            a0 = kb::ext_add(a0, kb::ext_mul(r0, q0));
            if (x + 1 < col_end) {
                ah = kb::ext_add(ah, kb::ext_mul(kb::ext_add(r0, jg_row_eq(J, x + 1 - J.prefix[c])), kb::ext_add(q0, q1)));
            } else {
                Ext jsum = kb::ext_mul(ld_ext(J.col_eq, c), r0);
                if (x + 1 < n_out) {                      // x + 1 opens a later non-empty column
                    uint32_t c1 = c + 1;
                    while (J.prefix[c1 + 1] <= x + 1) c1++;
                    jsum = kb::ext_add(jsum, kb::ext_mul(ld_ext(J.col_eq, c1), jg_row_eq(J, x + 1 - J.prefix[c1])));
                }
                eh = kb::ext_add(eh, kb::ext_mul(jsum, kb::ext_add(q0, q1)));
            }
        }
        if (have) {
            const Ext w = ld_ext(J.col_eq, c);
            e0 = kb::ext_add(e0, kb::ext_mul(w, a0));
            eh = kb::ext_add(eh, kb::ext_mul(w, ah));
        }
    }
    jg_finish(e0, eh, tail);
}

// j_out[x] = eq_col[c] * eq_row_r[x - prefix_r[c]] for the n entries of level r (leaving the factored form)
__global__ __launch_bounds__(256) void jg_materialize_j(JgJ J, uint32_t n, Ext* __restrict__ j_out) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= n) return;
    const uint32_t c = jg_find_col(J, x);
    st_ext(j_out, x, jg_j_at(J, c, x - J.prefix[c]));
}

// ---- later folds: ext tables with n_in live entries -> n_out = ceil(n_in / 2), fused with the sums
__global__ __launch_bounds__(256) void jg_fold_sum(const Ext* __restrict__ q_in, const Ext* __restrict__ j_in, uint32_t n_in,
                                                   Ext alpha, uint32_t n_out, Ext* __restrict__ q_out, Ext* __restrict__ j_out,
                                                   JgTail tail) {
    Ext e0 = kb::ext_zero(), eh = kb::ext_zero();
    const uint32_t n_thr = (n_out + 1) / 2;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_thr; k += gridDim.x * blockDim.x) {
        Ext q[4], j[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const uint32_t x = 4 * k + v;
            q[v] = x < n_in ? ld_ext(q_in, x) : kb::ext_zero();
            j[v] = x < n_in ? ld_ext(j_in, x) : kb::ext_zero();
        }
        Ext qo[2], jo[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            qo[h] = fold_ext(q[2 * h], q[2 * h + 1], alpha);
            jo[h] = fold_ext(j[2 * h], j[2 * h + 1], alpha);
            if (2 * k + h < n_out) { st_ext(q_out, 2 * k + h, qo[h]); st_ext(j_out, 2 * k + h, jo[h]); }
        }
        e0 = kb::ext_add(e0, kb::ext_mul(jo[0], qo[0]));
        eh = kb::ext_add(eh, kb::ext_mul(kb::ext_add(jo[0], jo[1]), kb::ext_add(qo[0], qo[1])));
    }
    jg_finish(e0, eh, tail);
}

// The same fold with J still FACTORED on the way in (the first round after the factored levels): a lane computes the j of its
// four entries from eq_col x eq_row (one column search, then a walk) instead of reading a table that jg_materialize_j would
// have had to write at full size first — 16 B written and 16 B read per entry of the largest generic level saved.
__global__ __launch_bounds__(256) void jg_fold_sum_fj(const Ext* __restrict__ q_in, JgJ J, uint32_t n_in, Ext alpha, uint32_t n_out,
                                                      Ext* __restrict__ q_out, Ext* __restrict__ j_out, JgTail tail) {
    Ext e0 = kb::ext_zero(), eh = kb::ext_zero();
    const uint32_t n_thr = (n_out + 1) / 2;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_thr; k += gridDim.x * blockDim.x) {
        Ext q[4], j[4];
        uint32_t c = 0;
        bool have = false;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const uint32_t x = 4 * k + v;
            if (x >= n_in) { q[v] = kb::ext_zero(); j[v] = kb::ext_zero(); continue; }
            q[v] = ld_ext(q_in, x);
            if (!have) { c = jg_find_col(J, x); have = true; }
            else while (J.prefix[c + 1] <= x) c++;
            j[v] = jg_j_at(J, c, x - J.prefix[c]);
        }
        Ext qo[2], jo[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            qo[h] = fold_ext(q[2 * h], q[2 * h + 1], alpha);
            jo[h] = fold_ext(j[2 * h], j[2 * h + 1], alpha);
            if (2 * k + h < n_out) { st_ext(q_out, 2 * k + h, qo[h]); st_ext(j_out, 2 * k + h, jo[h]); }
        }
        e0 = kb::ext_add(e0, kb::ext_mul(jo[0], qo[0]));
        eh = kb::ext_add(eh, kb::ext_mul(kb::ext_add(jo[0], jo[1]), kb::ext_add(qo[0], qo[1])));
    }
    jg_finish(e0, eh, tail);
}

// ================================================================ jagged-eval sumcheck
// Transfer matrices of the branching program. A layer reads (row bit, index bit, curr-prefix bit,
// next-prefix bit); with the first two bound to z_row / z_trace values the layer acts on the 4 memory
// states as a 4x4 matrix that is multilinear in (curr, next): mats[layer][2 cb + nb][m][m'].
struct JeCols {                     // condensed columns (runs of equal (t_c, t_{c+1}) merged)
    const uint32_t* t;              // t_c
    const uint32_t* u;              // t_{c+1}
    const Ext* zcol;                // summed eq(z_col, c) of the run
    uint32_t n;
};

__device__ __forceinline__ void matvec(const Ext* __restrict__ A, const Ext (&s)[4], Ext (&out)[4]) {   // out = A s
#pragma unroll
    for (int m = 0; m < 4; m++) {
        Ext acc = kb::ext_mul(ld_ext(A, 4 * m), s[0]);
#pragma unroll
        for (int k = 1; k < 4; k++) acc = kb::ext_add(acc, kb::ext_mul(ld_ext(A, 4 * m + k), s[k]));
        out[m] = acc;
    }
}
__device__ __forceinline__ void vecmat(const Ext (&w)[4], const Ext* __restrict__ A, Ext (&out)[4]) {   // out = w^T A
#pragma unroll
    for (int k = 0; k < 4; k++) {
        Ext acc = kb::ext_mul(w[0], ld_ext(A, k));
#pragma unroll
        for (int m = 1; m < 4; m++) acc = kb::ext_add(acc, kb::ext_mul(w[m], ld_ext(A, 4 * m + k)));
        out[k] = acc;
    }
}
__device__ __forceinline__ Ext dot4(const Ext (&a)[4], const Ext (&b)[4]) {
    Ext acc = kb::ext_mul(a[0], b[0]);
#pragma unroll
    for (int k = 1; k < 4; k++) acc = kb::ext_add(acc, kb::ext_mul(a[k], b[k]));
    return acc;
}

// suffix[layer][k] = prod_{l >= layer} M_l(column k) e_success, for layer = D-1 .. 0.
// PAIR = true: matrices indexed by (t bit, u bit) (mats has 4 per layer); false: by the t bit only (2 per layer).
// Also accumulates sum_k zcol[k] * suffix[0][k][initial state] into out8[0..3] (full J evaluation).
template <bool PAIR>
__global__ __launch_bounds__(256) void je_suffix_kernel(JeCols C, const Ext* __restrict__ mats, int D, Ext* __restrict__ suffix,
                                                        JgTail tail) {
    Ext total = kb::ext_zero();
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < C.n; k += gridDim.x * blockDim.x) {
        const uint32_t t = C.t[k], u = C.u[k];
        Ext s[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_one(), kb::ext_zero()};   // success = {carry 0, comparison 1}
        for (int layer = D - 1; layer >= 0; layer--) {
            const uint32_t cb = (t >> layer) & 1u, nb = (u >> layer) & 1u;
            const Ext* A = PAIR ? mats + ((size_t)layer * 4 + cb * 2 + nb) * 16 : mats + ((size_t)layer * 2 + cb) * 16;
            Ext o[4];
            matvec(A, s, o);
#pragma unroll
            for (int m = 0; m < 4; m++) { s[m] = o[m]; st_ext(suffix, ((uint32_t)layer * C.n + k) * 4 + m, o[m]); }
        }
        total = kb::ext_add(total, kb::ext_mul(ld_ext(C.zcol, k), s[0]));
    }
    jg_finish(total, kb::ext_zero(), tail);
}

// One round of the first half (variable = bit r of t_{c+1}). state[k] = {v0[4], v1[4]} of the previous
// round, inter[k] = eq accumulator. first: r == 0.
__global__ __launch_bounds__(256) void je_phase1_round(JeCols C, const Ext* __restrict__ mats, const Ext* __restrict__ suffix, int D,
                                                       int r, Ext alpha_prev, Ext half, Ext* __restrict__ state,
                                                       Ext* __restrict__ inter, JgTail tail) {
    Ext y0 = kb::ext_zero(), yh = kb::ext_zero();
    const Ext one = kb::ext_one();
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < C.n; k += gridDim.x * blockDim.x) {
        const uint32_t t = C.t[k], u = C.u[k];
        Ext w[4], it;
        if (r == 0) {
            w[0] = one; w[1] = w[2] = w[3] = kb::ext_zero();     // initial state = {carry 0, comparison 0}
            it = one;
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) w[m] = fold_ext(ld_ext(state, 8 * k + m), ld_ext(state, 8 * k + 4 + m), alpha_prev);
            const bool x = (u >> (r - 1)) & 1u;
            it = kb::ext_mul(ld_ext(inter, k), x ? alpha_prev : kb::ext_sub(one, alpha_prev));
        }
        st_ext(inter, k, it);
        const uint32_t cb = (t >> r) & 1u;
        Ext v0[4], v1[4], s[4];
        vecmat(w, mats + ((size_t)r * 4 + cb * 2 + 0) * 16, v0);
        vecmat(w, mats + ((size_t)r * 4 + cb * 2 + 1) * 16, v1);
#pragma unroll
        for (int m = 0; m < 4; m++) { st_ext(state, 8 * k + m, v0[m]); st_ext(state, 8 * k + 4 + m, v1[m]); }
        if (r + 1 < D)
#pragma unroll
            for (int m = 0; m < 4; m++) s[m] = ld_ext(suffix, ((uint32_t)(r + 1) * C.n + k) * 4 + m);
        else { s[0] = s[1] = s[3] = kb::ext_zero(); s[2] = one; }
        const Ext bp0 = dot4(v0, s), bp1 = dot4(v1, s);
        const Ext f = kb::ext_mul(ld_ext(C.zcol, k), it);
        if (!((u >> r) & 1u)) y0 = kb::ext_add(y0, kb::ext_mul(f, bp0));
        yh = kb::ext_add(yh, kb::ext_mul(kb::ext_mul(f, half), kb::ext_mul(kb::ext_add(bp0, bp1), half)));
    }
    jg_finish(y0, yh, tail);
}

// One round of the second half (variable = bit rp of t_c; every bit of t_{c+1} is bound). V0 / V1: the
// column-independent bound part times the two lambda-endpoint matrices (host). prev_bit_of_u: the
// previously bound variable was bit D-1 of u (rp == 0) or bit rp-1 of t.
struct JeV { Ext v[8]; };          // the two prefix row vectors of a phase-2 round, passed by value (128 B of kernarg)
__global__ __launch_bounds__(256) void je_phase2_round(JeCols C, const Ext* __restrict__ suffix2, int D, int rp, Ext alpha_prev,
                                                       Ext half, JeV V, Ext* __restrict__ inter,
                                                       JgTail tail) {
    Ext y0 = kb::ext_zero(), yh = kb::ext_zero();
    const Ext one = kb::ext_one();
    Ext v0[4], v1[4];
#pragma unroll
    for (int m = 0; m < 4; m++) { v0[m] = V.v[m]; v1[m] = V.v[4 + m]; }
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < C.n; k += gridDim.x * blockDim.x) {
        const uint32_t t = C.t[k], u = C.u[k];
        const bool x = rp == 0 ? ((u >> (D - 1)) & 1u) : ((t >> (rp - 1)) & 1u);
        const Ext it = kb::ext_mul(ld_ext(inter, k), x ? alpha_prev : kb::ext_sub(one, alpha_prev));
        st_ext(inter, k, it);
        Ext s[4];
        if (rp + 1 < D)
#pragma unroll
            for (int m = 0; m < 4; m++) s[m] = ld_ext(suffix2, ((uint32_t)(rp + 1) * C.n + k) * 4 + m);
        else { s[0] = s[1] = s[3] = kb::ext_zero(); s[2] = one; }
        const Ext bp0 = dot4(v0, s), bp1 = dot4(v1, s);
        const Ext f = kb::ext_mul(ld_ext(C.zcol, k), it);
        if (!((t >> rp) & 1u)) y0 = kb::ext_add(y0, kb::ext_mul(f, bp0));
        yh = kb::ext_add(yh, kb::ext_mul(kb::ext_mul(f, half), kb::ext_mul(kb::ext_add(bp0, bp1), half)));
    }
    jg_finish(y0, yh, tail);
}

// ================================================================ host driver
namespace {

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

Ext eval_ext_mle_host(const std::vector<Ext>& vals, const std::vector<Ext>& pt) {
    const std::vector<Ext> eq = partial_lagrange_host(pt);
    Ext acc = kb::ext_zero();
    for (size_t i = 0; i < vals.size() && i < eq.size(); i++) acc = acc + eq[i] * vals[i];
    return acc;
}

struct Bytes {                     // bincode writer over the caller's proof buffer
    uint8_t* p = nullptr;
    size_t cap = 0, n = 0;
    bool overflow = false;
    void raw(const void* src, size_t k) { if (n + k > cap) { overflow = true; return; } memcpy(p + n, src, k); n += k; }
    void u64(uint64_t v) { raw(&v, 8); }               // little-endian host
    void felt(uint32_t m) { const uint32_t v = kb::from_monty(m); raw(&v, 4); }
    void ext(const Ext& e) { for (int k = 0; k < 4; k++) felt(e.c[k]); }
};

struct Sumcheck {                  // PartialSumcheckProof<EF> (sumcheck/src/proof.rs:L10-L14)
    std::vector<std::array<Ext, 3>> polys;
    Ext claimed_sum, eval;
    std::vector<Ext> point;
    void write(Bytes& w) const {
        w.u64(polys.size());
        for (auto& p : polys) { w.u64(3); for (auto& c : p) w.ext(c); }
        w.ext(claimed_sum);
        w.u64(point.size());
        for (auto& x : point) w.ext(x);
        w.ext(eval);
    }
};

Ext poly_eval(const std::array<Ext, 3>& c, const Ext& x) { return (c[2] * x + c[1]) * x + c[0]; }

// degree-2 interpolation through y(0), y(1/2), y(1) given y0, H = 4 y(1/2), y1: c2 = 2 (y0 + y1) - H
std::array<Ext, 3> interpolate(const Ext& y0, const Ext& h4, const Ext& y1) {
    const Ext s = y0 + y1;
    const Ext c2 = s + s - h4;
    return {y0, y1 - y0 - c2, c2};
}

Ext observe_and_sample(sp1hip_challenger_t* ch, const std::array<Ext, 3>& poly) {
    for (auto& c : poly) for (int k = 0; k < 4; k++) challenger_observe(ch, c.c[k]);
    return challenger_sample_ext(ch);
}

struct Scratch {                    // device scratch shared by all rounds of one proof
    DeviceBuf partials;
    RoundSyncHost rsync;
    Mailbox mb;
    PinnedStage stage;
    uint32_t h_out[8];
    hipStream_t s;
    static constexpr uint32_t MAX_BLOCKS = 2048;
    int init(hipStream_t stream) {
        s = stream;
        SP1HIP_TRY(partials.alloc((size_t)MAX_BLOCKS * 128, s));      // 8 extension sums per workgroup at most (jg_round01_tables)
        SP1HIP_TRY(stage.init(s));
        SP1HIP_TRY(rsync.init(s));
        return mb.init(s);
    }
    // argument of the round kernel about to be launched (one per launch: it advances the sequence number finish() waits for)
    JgTail tail() { const RoundSync rs = rsync.next(); return JgTail{partials.u32(), rs, rsync.seq}; }
    static uint32_t blocks_for(uint64_t threads) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>((threads + 255) / 256, 1), MAX_BLOCKS); }
    // reduce `nb` block partials and bring the two ext sums to the host
    int finish8(Ext (&out)[8]) {
        uint32_t h[32];
        SP1HIP_TRY(rsync.wait(h, 32));
        memcpy(out, h, 128);
        return SP1HIP_SUCCESS;
    }
    int finish(uint32_t nb, Ext* a, Ext* b) {
        (void)nb;
        SP1HIP_TRY(rsync.wait(h_out, 8));
        memcpy(a, h_out, 16);
        memcpy(b, h_out + 4, 16);
        return SP1HIP_SUCCESS;
    }
};

// ---- branching-program layer matrices (poly.rs:L120-L160): out[layer][2 cb + nb][m][m']
void build_layer_matrices(const std::vector<Ext>& z_row, const std::vector<Ext>& z_index, int D, std::vector<Ext>* out) {
    auto lsb = [](const std::vector<Ext>& p, size_t i) { return p.size() <= i ? kb::ext_zero() : p[p.size() - 1 - i]; };
    out->assign((size_t)D * 4 * 16, kb::ext_zero());
    const Ext one = kb::ext_one();
    for (int layer = 0; layer < D; layer++) {
        const Ext zr = lsb(z_row, layer), zi = lsb(z_index, layer);
        const Ext er[2] = {one - zr, zr}, ei[2] = {one - zi, zi};
        for (int cb = 0; cb < 2; cb++)
            for (int nb = 0; nb < 2; nb++)
                for (int m = 0; m < 4; m++) {
                    const int carry = m & 1, cmp = m >> 1;
                    for (int rb = 0; rb < 2; rb++)
                        for (int ib = 0; ib < 2; ib++) {
                            const int sum = rb + carry + cb;
                            if (ib != (sum & 1)) continue;                       // fail transition
                            const int new_cmp = ib == nb ? cmp : nb;
                            const int target = (sum >> 1) + 2 * new_cmp;
                            Ext& e = (*out)[(((size_t)layer * 4 + cb * 2 + nb) * 4 + m) * 4 + target];
                            e = e + er[rb] * ei[ib];
                        }
                }
    }
}

int upload(DeviceBuf& buf, const void* src, size_t bytes, hipStream_t s, PinnedStage& stage) {
    SP1HIP_TRY(buf.alloc(std::max<size_t>(bytes, 16), s));
    return stage.upload(buf.p, src, bytes);
}

// JaggedEvalSumcheckProver::prove_jagged_evaluation. prefix: #columns + 1 dense prefix sums.
int jagged_eval_prove(const std::vector<uint32_t>& prefix, int log_m, const std::vector<Ext>& z_row, const std::vector<Ext>& z_col,
                      const std::vector<Ext>& z_trace, sp1hip_challenger_t* ch, Scratch& sc, Sumcheck* proof) {
    hipStream_t s = sc.s;
    const int D = log_m + 1, dim = 2 * D;
    // condensed (t_c, t_{c+1}) runs with their summed eq(z_col, .) weights (sumcheck_poly.rs:L106-L121)
    const std::vector<Ext> col_eq = partial_lagrange_host(z_col);
    std::vector<uint32_t> ts, us;
    std::vector<Ext> zc;
    for (size_t c = 0; c + 1 < prefix.size(); c++) {
        if (!ts.empty() && ts.back() == prefix[c] && us.back() == prefix[c + 1]) zc.back() = zc.back() + col_eq[c];
        else { ts.push_back(prefix[c]); us.push_back(prefix[c + 1]); zc.push_back(col_eq[c]); }
    }
    const uint32_t n = (uint32_t)ts.size();
    DeviceBuf d_t, d_u, d_zc, d_mats, d_suffix, d_state, d_inter;
    SP1HIP_TRY(upload(d_t, ts.data(), n * 4, s, sc.stage));
    SP1HIP_TRY(upload(d_u, us.data(), n * 4, s, sc.stage));
    SP1HIP_TRY(upload(d_zc, zc.data(), (size_t)n * 16, s, sc.stage));
    std::vector<Ext> mats;
    build_layer_matrices(z_row, z_trace, D, &mats);
    SP1HIP_TRY(upload(d_mats, mats.data(), mats.size() * 16, s, sc.stage));
    SP1HIP_TRY(d_suffix.alloc((size_t)D * n * 64, s));
    SP1HIP_TRY(d_state.alloc((size_t)n * 128, s));
    SP1HIP_TRY(d_inter.alloc((size_t)n * 16, s));
    const JeCols C{d_t.u32(), d_u.u32(), (const Ext*)d_zc.p, n};
    const uint32_t nb = Scratch::blocks_for(n);

    // all-boolean suffixes; their layer-0 entry gives the claimed J(z_trace) (full_jagged_little_polynomial_evaluation)
    hipLaunchKernelGGL(je_suffix_kernel<true>, dim3(nb), dim3(256), 0, s, C, (const Ext*)d_mats.p, D, (Ext*)d_suffix.p, sc.tail());
    SP1HIP_LAUNCH_CHECK();
    Ext expected_sum, unused;
    SP1HIP_TRY(sc.finish(nb, &expected_sum, &unused));
    for (int k = 0; k < 4; k++) challenger_observe(ch, expected_sum.c[k]);

    proof->claimed_sum = expected_sum;
    const Ext half = kb::ext_inv(ext_c(2));
    Ext claim = expected_sum, alpha = kb::ext_zero();
    std::vector<Ext> alphas;                       // sampling order
    std::array<Ext, 3> poly{};
    for (int r = 0; r < D; r++) {
        hipLaunchKernelGGL(je_phase1_round, dim3(nb), dim3(256), 0, s, C, (const Ext*)d_mats.p, (const Ext*)d_suffix.p, D, r, alpha,
                           half, (Ext*)d_state.p, (Ext*)d_inter.p, sc.tail());
        SP1HIP_LAUNCH_CHECK();
        Ext y0, yh;
        SP1HIP_TRY(sc.finish(nb, &y0, &yh));
        poly = interpolate(y0, yh + yh + yh + yh, claim - y0);
        proof->polys.push_back(poly);
        alpha = observe_and_sample(ch, poly);
        alphas.push_back(alpha);
        claim = poly_eval(poly, alpha);
    }
    // second half: every bit of t_{c+1} is bound to alphas[0..D); B[layer][cb] = A_layer(cb, alpha_layer)
    std::vector<Ext> B((size_t)D * 2 * 16);
    for (int layer = 0; layer < D; layer++)
        for (int cb = 0; cb < 2; cb++)
            for (int e = 0; e < 16; e++) {
                const Ext a0 = mats[((size_t)layer * 4 + cb * 2 + 0) * 16 + e], a1 = mats[((size_t)layer * 4 + cb * 2 + 1) * 16 + e];
                B[((size_t)layer * 2 + cb) * 16 + e] = a0 + alphas[layer] * (a1 - a0);
            }
    DeviceBuf d_B, d_suffix2;
    SP1HIP_TRY(upload(d_B, B.data(), B.size() * 16, s, sc.stage));
    SP1HIP_TRY(d_suffix2.alloc((size_t)D * n * 64, s));
    hipLaunchKernelGGL(je_suffix_kernel<false>, dim3(nb), dim3(256), 0, s, C, (const Ext*)d_B.p, D, (Ext*)d_suffix2.p, sc.tail());
    SP1HIP_LAUNCH_CHECK();
    Ext W[4] = {kb::ext_one(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};     // e_initial^T
    for (int rp = 0; rp < D; rp++) {
        Ext V[8];
        for (int b = 0; b < 2; b++)
            for (int k = 0; k < 4; k++) {
                Ext acc = kb::ext_zero();
                for (int m = 0; m < 4; m++) acc = acc + W[m] * B[((size_t)rp * 2 + b) * 16 + 4 * m + k];
                V[4 * b + k] = acc;
            }
        JeV Varg;
        memcpy(Varg.v, V, sizeof V);
        hipLaunchKernelGGL(je_phase2_round, dim3(nb), dim3(256), 0, s, C, (const Ext*)d_suffix2.p, D, rp, alpha, half,
                           Varg, (Ext*)d_inter.p, sc.tail());
        SP1HIP_LAUNCH_CHECK();
        Ext y0, yh;
        SP1HIP_TRY(sc.finish(nb, &y0, &yh));
        poly = interpolate(y0, yh + yh + yh + yh, claim - y0);
        proof->polys.push_back(poly);
        alpha = observe_and_sample(ch, poly);
        alphas.push_back(alpha);
        claim = poly_eval(poly, alpha);
        for (int k = 0; k < 4; k++) W[k] = V[k] + alpha * (V[4 + k] - V[k]);
    }
    (void)dim;
    proof->point.assign(alphas.rbegin(), alphas.rend());
    proof->eval = claim;
    return SP1HIP_SUCCESS;
}

}  // namespace

// bincode(JaggedPcsProof) size: BaseFold proof + batch evaluations + two sumchecks + counts + commitments + tail
size_t jagged_proof_size(int lsh, const std::vector<uint32_t>& round_widths, const std::vector<size_t>& tables_per_round,
                         uint64_t total_area, sp1hip_fri_config_t config) {
    const int n_rounds = (int)round_widths.size();
    const int log_m = log2_ceil(total_area), D = log_m + 1;
    size_t need = sp1hip_basefold_proof_size(lsh, round_widths.data(), n_rounds, config);
    need += 8;
    for (uint32_t w : round_widths) need += 8 + (size_t)w * 16 + 16;
    need += 8 + (size_t)log_m * (8 + 48) + 16 + 8 + (size_t)log_m * 16 + 16;
    need += 8 + (size_t)2 * D * (8 + 48) + 16 + 8 + (size_t)2 * D * 16 + 16;
    need += 8;
    for (size_t t : tables_per_round) need += 8 + t * 16;
    need += 8 + (size_t)n_rounds * 32 + 16 + 8 + 8;
    return need;
}
}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_jagged_prove(const sp1hip_ext_t* h_z_row, int max_log_row_count, sp1hip_stacked_data_t* const* rounds, int n_rounds,
                        const sp1hip_ext_t* h_claims, const size_t* claims_per_round, sp1hip_fri_config_t config,
                        sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(h_z_row && rounds && n_rounds > 0 && n_rounds <= 8 && claims_per_round && challenger && proof_len, "bad argument");
    SP1HIP_REQUIRE(max_log_row_count >= 0 && max_log_row_count <= 30, "max_log_row_count out of range");
    hipStream_t s = S(stream);
    // SP1HIP_JG_TIMING=1: host wall time of the call's parts on stderr
    const bool jg_timing = [] { const char* e = getenv("SP1HIP_JG_TIMING"); return e && e[0] == '1'; }();
    std::chrono::steady_clock::time_point jg_t[7];
    jg_t[0] = std::chrono::steady_clock::now();
    struct StageMarks {                                   // roctx sub-ranges of the evaluation proof (rocprofv3 --marker-trace)
        bool open = false;
        void next(const char* name) { if (open) roctx_pop(); roctx_push(name); open = true; }
        ~StageMarks() { if (open) roctx_pop(); }
    } marks;
    marks.next("jagged_setup");
    const int lsh = rounds[0]->log_stacking_height;
    SP1HIP_REQUIRE(lsh >= 1, "log_stacking_height must be at least 1");
    uint64_t total_cols = 0, total_area = 0, n_claims = 0;
    std::vector<uint32_t> round_widths;
    for (int r = 0; r < n_rounds; r++) {
        const sp1hip_stacked_data_s* d = rounds[r];
        SP1HIP_REQUIRE(d && d->jagged, "round data did not come from sp1hip_jagged_commit");
        if (d->stream != s) rounds[r]->foreign_use = true;
        SP1HIP_REQUIRE(d->log_stacking_height == lsh && d->max_log_row_count == max_log_row_count, "rounds disagree on parameters");
        SP1HIP_REQUIRE(d->area > 0, "a commitment round without any table data cannot be opened");
        uint64_t expect = 0;
        for (size_t t = 0; t + 2 < d->column_counts.size(); t++) expect += d->column_counts[t];
        SP1HIP_REQUIRE(claims_per_round[r] == expect, "claims_per_round does not match the committed tables");
        for (uint64_t c : d->column_counts) total_cols += c;
        total_area += d->padded;
        n_claims += claims_per_round[r];
        round_widths.push_back((uint32_t)(d->padded >> lsh));
    }
    SP1HIP_REQUIRE(h_claims || n_claims == 0, "null claims");
    SP1HIP_REQUIRE(total_area < ((uint64_t)1 << 30), "dense area must stay below 2^30 (jagged verifier bound)");
    const int log_m = log2_ceil(total_area);
    SP1HIP_REQUIRE(log_m >= lsh, "internal: area smaller than one stacked column");
    const int num_col_variables = log2_ceil(total_cols);
    std::vector<size_t> tables_per_round;
    for (int r = 0; r < n_rounds; r++) tables_per_round.push_back(rounds[r]->row_counts.size());
    const size_t need = jagged_proof_size(lsh, round_widths, tables_per_round, total_area, config);
    if (!h_proof || *proof_len < need) {
        *proof_len = need;
        set_error("sp1hip_jagged_prove: proof buffer too small, need %zu bytes", need);
        return SP1HIP_ERROR_BUFFER_TOO_SMALL;
    }
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));

    // work on a copy of the transcript; commit it only on success
    sp1hip_challenger_t* ch = nullptr;
    SP1HIP_TRY(sp1hip_challenger_clone(challenger, &ch));
    struct ChGuard { sp1hip_challenger_t* c; ~ChGuard() { sp1hip_challenger_free(c); } } guard{ch};

    std::vector<Ext> z_row(max_log_row_count), z_col(num_col_variables);
    memcpy(z_row.data(), h_z_row, (size_t)max_log_row_count * 16);
    for (auto& z : z_col) z = challenger_sample_ext(ch);

    // column claims with the zero claims of the padding columns, padded to a power of two (prover.rs:L187-L212, L268-L271)
    std::vector<Ext> column_claims;
    {
        size_t off = 0;
        for (int r = 0; r < n_rounds; r++) {
            for (size_t i = 0; i < claims_per_round[r]; i++) { Ext e; memcpy(&e, &h_claims[off + i], 16); column_claims.push_back(e); }
            off += claims_per_round[r];
            column_claims.insert(column_claims.end(), rounds[r]->padding_column_count, kb::ext_zero());
        }
    }
    SP1HIP_REQUIRE(column_claims.size() == total_cols, "internal: column claim count");
    const Ext sumcheck_claim = eval_ext_mle_host(column_claims, z_col);

    // dense prefix sums, one entry per column (JaggedLittlePolynomialProverParams::new)
    std::vector<uint32_t> prefix;
    {
        uint64_t acc = 0;
        for (int r = 0; r < n_rounds; r++)
            for (size_t t = 0; t < rounds[r]->row_counts.size(); t++)
                for (uint64_t c = 0; c < rounds[r]->column_counts[t]; c++) { prefix.push_back((uint32_t)acc); acc += rounds[r]->row_counts[t]; }
        prefix.push_back((uint32_t)acc);
        SP1HIP_REQUIRE(acc == total_area, "internal: prefix sums do not cover the dense area");
    }
    const uint32_t ncols = (uint32_t)prefix.size() - 1;
    const std::vector<Ext> col_eq = partial_lagrange_host(z_col);

    Scratch sc;
    SP1HIP_TRY(sc.init(s));
    DeviceBuf d_prefix, d_col_eq, d_row_eq;
    SP1HIP_TRY(upload(d_prefix, prefix.data(), prefix.size() * 4, s, sc.stage));
    SP1HIP_TRY(upload(d_col_eq, col_eq.data(), col_eq.size() * 16, s, sc.stage));
    SP1HIP_TRY(d_row_eq.alloc(((size_t)16) << max_log_row_count, s));
    SP1HIP_TRY(sp1hip_partial_lagrange(h_z_row, max_log_row_count, d_row_eq.u32(), stream));
    JgSegs segs{};
    segs.n = n_rounds;
    {
        uint64_t start = 0;
        for (int r = 0; r < n_rounds; r++) {
            segs.s[r] = JgSeg{(const uint32_t*)rounds[r]->d_dense, (uint32_t)start, (uint32_t)rounds[r]->padded};
            start += rounds[r]->padded;
        }
        segs.total = (uint32_t)start;
    }
    const JgJ J{d_prefix.u32(), ncols, (const Ext*)d_col_eq.p, d_row_eq.u32(), 1u << max_log_row_count};
    const uint32_t T = segs.total;

    // ---- jagged sumcheck: log_m rounds
    marks.next("jagged_sumcheck");
    Sumcheck sumcheck;
    sumcheck.claimed_sum = sumcheck_claim;
    std::vector<Ext> alphas;
    Ext claim = sumcheck_claim, alpha = kb::ext_zero(), q_eval = kb::ext_zero(), j_eval = kb::ext_zero();
    std::array<Ext, 3> poly{};
    DeviceBuf tabs[4];                 // q/j ping-pong: q of odd levels in tabs[0], of even levels in tabs[2]; j behind each
    const uint32_t n1 = (T + 1) / 2;
    // levels 1 .. rf keep J factored (see jg_foldf_sum): every column must start at a multiple of 2^level
    int rf = 0;
    {
        uint32_t g = T;
        for (uint32_t pfx : prefix) g |= pfx;
        rf = g ? __builtin_ctz(g) : 0;
        rf = std::min(rf, std::min(log_m - 2, max_log_row_count - 1));
        if (const char* e = getenv("SP1HIP_JAGGED_FACTORED")) if (e[0] == '0') rf = 0;     // A/B switch
        if (rf < 2) rf = 0;            // a single factored level is not worth the extra table
    }
    bool j_materialised = rf == 0;     // does tabs[cur + 1] hold the j table of the current level?
    if (rf == 0) {
        SP1HIP_TRY(tabs[1].alloc((size_t)std::max<uint32_t>(n1, 1) * 16, s));
        SP1HIP_TRY(tabs[3].alloc((size_t)std::max<uint32_t>((n1 + 1) / 2, 1) * 16, s));
    }
    DeviceBuf row_eq_lv[2], d_prefix_lv;       // folded row tables (ping-pong) and the shifted prefix sums of a level
    const uint32_t* row_eq_cur = d_row_eq.u32();
    uint32_t row_len_cur = 1u << max_log_row_count;
    if (rf) {
        SP1HIP_TRY(row_eq_lv[0].alloc(((size_t)16) << (max_log_row_count - 1), s));
        SP1HIP_TRY(row_eq_lv[1].alloc(((size_t)16) << (max_log_row_count - 1), s));
        SP1HIP_TRY(d_prefix_lv.alloc(prefix.size() * 4 * (size_t)(rf + 1), s));
        std::vector<uint32_t> all_lv(prefix.size() * (size_t)(rf + 1));      // the shifted prefix sums of every level: one upload
        for (int lv = 0; lv <= rf; lv++)
            for (size_t c = 0; c < prefix.size(); c++) all_lv[(size_t)lv * prefix.size() + c] = prefix[c] >> lv;
        SP1HIP_TRY(sc.stage.upload(d_prefix_lv.p, all_lv.data(), all_lv.size() * 4));
    }
    // J of level `lv` (1 <= lv <= rf): folds the row table once more with the challenge that produced the level
    auto level_J = [&](int lv, const Ext& a_prev, JgJ* out) -> int {
        uint32_t* dst = row_eq_lv[lv & 1].u32();
        hipLaunchKernelGGL(jg_fold_row_eq, dim3((row_len_cur / 2 + 255) / 256), dim3(256), 0, s, row_eq_cur, row_len_cur, a_prev, dst);
        SP1HIP_LAUNCH_CHECK();
        row_eq_cur = dst;
        row_len_cur >>= 1;
        uint32_t* d_pl = d_prefix_lv.u32() + (size_t)lv * prefix.size();
        *out = JgJ{d_pl, ncols, (const Ext*)d_col_eq.p, row_eq_cur, row_len_cur};
        return SP1HIP_SUCCESS;
    };
    // table-major descriptors for round 0 (only when every column start is even, i.e. dense pairs never straddle columns)
    std::vector<JgTab> tabs0;
    uint32_t n_tiles0 = 0, n_tiles1 = 0, n_tiles2 = 0;
    bool tables_mult4 = false;         // every table height a multiple of 4: the first fold can go table-major too
    bool skip_level1 = false;          // ... of 8: rounds 1 and 2 both fold from the base words, level 1 is never stored
    std::vector<std::pair<uint64_t, uint64_t>> zero_tails;      // dense index ranges holding only padding zeros
    DeviceBuf d_tabs0, d_col_eq3;
    {
        uint32_t g = T;
        for (uint32_t pfx : prefix) g |= pfx;
        const char* e = getenv("SP1HIP_JAGGED_FACTORED");
        if ((g & 1u) == 0 && !(e && e[0] == '0')) {
            uint64_t seg_start = 0;
            uint32_t col = 0;
            for (int r = 0; r < n_rounds; r++) {
                const sp1hip_stacked_data_s* d = rounds[r];
                uint64_t off = 0;
                const size_t n_real = d->row_counts.size() - 2;           // the two padding tables hold zeros only
                for (size_t t = 0; t < d->row_counts.size(); t++) {
                    const uint32_t h = (uint32_t)d->row_counts[t], w = (uint32_t)d->column_counts[t];
                    if (t < n_real && h && w) {
                        // a lane of the table-major kernels owns a few rows of a table and walks its columns one after the other, so a
                        // wide, short table is a few waves with a long dependent loop: the 2,640 columns x 122k rows of a Keccak shard
                        // made jg_fold_tables<2> one 3.3 ms launch on 60 workgroups (0.7 ms on a core shard of the same area). Every
                        // quantity is a sum over columns, so a table enters as slices of at most `col_slice` columns — descriptors of
                        // their own, the kernels see narrower tables. SP1HIP_JAGGED_COL_SLICE=0: whole tables (A/B; same bytes).
                        static const uint32_t col_slice = [] { const char* e2 = getenv("SP1HIP_JAGGED_COL_SLICE"); return e2 ? (uint32_t)strtoul(e2, nullptr, 10) : 64u; }();
                        const uint32_t step = col_slice ? col_slice : w;
                        for (uint32_t c_lo = 0; c_lo < w; c_lo += step) {
                            const uint32_t wc = std::min(step, w - c_lo);
                            const uint64_t o = off + (uint64_t)c_lo * h;
                            tabs0.push_back(JgTab{(const uint32_t*)d->d_dense + o, h, col + c_lo, wc, n_tiles0, (uint32_t)(seg_start + o), n_tiles1, n_tiles2});
                            n_tiles0 += (h / 2 + JG_TAB_PAIRS - 1) / JG_TAB_PAIRS;
                            n_tiles1 += ((h + 3) / 4 + JG_TAB_PAIRS - 1) / JG_TAB_PAIRS;
                            n_tiles2 += ((h + 7) / 8 + JG_TAB_PAIRS - 1) / JG_TAB_PAIRS;
                        }
                    }
                    if (t == n_real) zero_tails.push_back({seg_start + off, seg_start + d->padded});
                    off += (uint64_t)h * w;
                    col += w;
                }
                seg_start += d->padded;
            }
            if (!tabs0.empty()) {
                SP1HIP_TRY(upload(d_tabs0, tabs0.data(), tabs0.size() * sizeof(JgTab), s, sc.stage));
                tables_mult4 = (g & 3u) == 0;
                skip_level1 = (g & 7u) == 0 && rf >= 2 && log_m >= 4;
                std::vector<Ext> col_eq3(col_eq.size());
                for (size_t c = 0; c < col_eq.size(); c++)
                    for (int q = 0; q < 4; q++) col_eq3[c].c[q] = kb::add(kb::dbl(col_eq[c].c[q]), col_eq[c].c[q]);
                SP1HIP_TRY(upload(d_col_eq3, col_eq3.data(), col_eq3.size() * 16, s, sc.stage));
            }
        }
    }
    // q of odd levels lives in tabs[0]: level 1 (T/2 entries), or level 3 when level 1 is never stored
    SP1HIP_TRY(tabs[0].alloc((size_t)std::max<uint32_t>(skip_level1 ? (T + 7) / 8 : n1, 1) * 16, s));
    SP1HIP_TRY(tabs[2].alloc((size_t)std::max<uint32_t>((n1 + 1) / 2, 1) * 16, s));
    uint32_t n_live = T;               // live entries of the current round's tables
    int cur = 0;                       // tabs[cur] (and tabs[cur + 1] once materialised) hold (q, j) of the current level
    jg_t[1] = std::chrono::steady_clock::now();
    // SP1HIP_JAGGED_LOOKAHEAD=0: rounds 0 and 1 as two passes over the base words (the A/B form; the GPU tests run both)
    const bool lookahead01 = skip_level1 && !tabs0.empty() && log_m >= 2 && !([] { const char* e = getenv("SP1HIP_JAGGED_LOOKAHEAD"); return e && e[0] == '0'; }());
    for (int round = 0; round < log_m; round++) {
        uint32_t nb;
        if (round == 0 && lookahead01) {
            // rounds 0 and 1 from ONE pass (jg_round01_tables); round 2 then folds from the base words with both challenges
            {
                ScopedTimer t("jagged_round0_sum", s);
                nb = Scratch::blocks_for((uint64_t)n_tiles1 * 256);
                hipLaunchKernelGGL(jg_round01_tables, dim3(nb), dim3(256), 0, s, (const JgTab*)d_tabs0.p, (uint32_t)tabs0.size(), n_tiles1, J,
                                   sc.tail());
                SP1HIP_LAUNCH_CHECK();
            }
            Ext S[8];
            SP1HIP_TRY(sc.finish8(S));
            // round 0
            const Ext y0 = S[0] + S[1], h0 = S[2] + S[3];
            poly = interpolate(y0, h0, claim - y0);
            sumcheck.polys.push_back(poly);
            alpha = observe_and_sample(ch, poly);
            alphas.push_back(alpha);
            claim = poly_eval(poly, alpha);
            // the row-eq table of level 1 (the state later levels fold from), as round 1 would have built it
            {
                JgJ J1;
                SP1HIP_TRY(level_J(1, alpha, &J1));
            }
            // round 1: y'(0) and H' are quadratics in alpha_0 through their values at 0, 1 and their leading coefficients
            const Ext a0 = alpha;
            const Ext c0 = S[0], c2 = (S[0] + S[4]) + (S[0] + S[4]) - S[2], c1 = S[4] - c0 - c2;
            const Ext d0 = S[5], d2 = S[7], d1 = S[6] - d0 - d2;
            const Ext y1 = c0 + (c1 + c2 * a0) * a0, h1 = d0 + (d1 + d2 * a0) * a0;
            poly = interpolate(y1, h1, claim - y1);
            sumcheck.polys.push_back(poly);
            alpha = observe_and_sample(ch, poly);
            alphas.push_back(alpha);
            claim = poly_eval(poly, alpha);
            n_live = (n_live + 1) / 2;
            cur = 0;
            round = 1;                                       // the loop continues with round 2
            continue;
        }
        if (round == 0) {
            ScopedTimer t("jagged_round0_sum", s);
            if (!tabs0.empty()) {
                nb = Scratch::blocks_for((uint64_t)n_tiles0 * 256);
                hipLaunchKernelGGL(jg_round0_tables, dim3(nb), dim3(256), 0, s, (const JgTab*)d_tabs0.p, (uint32_t)tabs0.size(), n_tiles0, J,
                                   sc.tail());
            } else {
                nb = Scratch::blocks_for((T / 2 + JG_ITERS - 1) / JG_ITERS);
                hipLaunchKernelGGL(jg_round0_sum, dim3(nb), dim3(256), 0, s, segs, J, T / 2, sc.tail());
            }
        } else if (round == 1) {
            ScopedTimer t("jagged_fold0_sum", s);
            const uint32_t n_out = (n_live + 1) / 2;
            nb = Scratch::blocks_for(((n_out + 1) / 2 + JG_ITERS - 1) / JG_ITERS);
            if (rf) {
                JgJ J1;                                    // eq_row_1 for this round's sums and for level 2
                SP1HIP_TRY(level_J(1, alpha, &J1));
                if (skip_level1) {                         // sums only; round 2 folds from the base words again
                    nb = Scratch::blocks_for((uint64_t)n_tiles1 * 256);
                    hipLaunchKernelGGL((jg_fold_tables<1, false>), dim3(nb), dim3(256), 0, s, (const JgTab*)d_tabs0.p, (uint32_t)tabs0.size(),
                                       n_tiles1, J1, (const Ext*)d_col_eq3.p, alpha, alpha, (Ext*)nullptr, sc.tail());
                } else if (tables_mult4) {
                    // the kernel writes the real tables only: the zero tail of every round (its padding tables) must
                    // read as zero in the next fold
                    for (auto& z : zero_tails)
                        if (z.second > z.first) SP1HIP_HIP(hipMemsetAsync((Ext*)tabs[0].p + z.first / 2, 0, (size_t)((z.second - z.first) / 2) * 16, s));
                    nb = Scratch::blocks_for((uint64_t)n_tiles1 * 256);
                    hipLaunchKernelGGL((jg_fold_tables<1, true>), dim3(nb), dim3(256), 0, s, (const JgTab*)d_tabs0.p, (uint32_t)tabs0.size(),
                                       n_tiles1, J1, (const Ext*)d_col_eq3.p, alpha, alpha, (Ext*)tabs[0].p, sc.tail());
                } else {
                    hipLaunchKernelGGL(jg_fold0_sum<false>, dim3(nb), dim3(256), 0, s, segs, J, alpha, n_out, (Ext*)tabs[0].p, (Ext*)nullptr,
                                       sc.tail());
                }
            } else {
                hipLaunchKernelGGL(jg_fold0_sum<true>, dim3(nb), dim3(256), 0, s, segs, J, alpha, n_out, (Ext*)tabs[0].p, (Ext*)tabs[1].p,
                                   sc.tail());
            }
            n_live = n_out;
            cur = 0;
        } else {
            ScopedTimer t("jagged_fold_sum", s);
            const uint32_t n_out = (n_live + 1) / 2;
            const int nxt = cur ^ 2;
            if (round == 2 && skip_level1) {               // level 2 straight from the base words (alpha0, alpha1)
                JgJ Jl;
                SP1HIP_TRY(level_J(2, alpha, &Jl));
                for (auto& z : zero_tails)
                    if (z.second > z.first) SP1HIP_HIP(hipMemsetAsync((Ext*)tabs[nxt].p + z.first / 4, 0, (size_t)((z.second - z.first) / 4) * 16, s));
                nb = Scratch::blocks_for((uint64_t)n_tiles2 * 256);
                hipLaunchKernelGGL((jg_fold_tables<2, true>), dim3(nb), dim3(256), 0, s, (const JgTab*)d_tabs0.p, (uint32_t)tabs0.size(),
                                   n_tiles2, Jl, (const Ext*)d_col_eq3.p, alphas[0], alpha, (Ext*)tabs[nxt].p, sc.tail());
            } else if (round <= rf) {                      // factored: level `round` from level `round - 1`
                JgJ Jl;
                SP1HIP_TRY(level_J(round, alpha, &Jl));
                nb = Scratch::blocks_for(((n_out + 1) / 2 + JG_ITERS - 1) / JG_ITERS);
                hipLaunchKernelGGL(jg_foldf_sum, dim3(nb), dim3(256), 0, s, (const Ext*)tabs[cur].p, n_live, Jl, alpha, n_out,
                                   (Ext*)tabs[nxt].p, sc.tail());
            } else {
                nb = Scratch::blocks_for((n_out + 1) / 2);
                if (!j_materialised) {                     // leave the factored form: this round computes j of level round - 1 as it loads
                    const JgJ Jp{d_prefix_lv.u32() + (size_t)(round - 1) * prefix.size(), ncols, (const Ext*)d_col_eq.p, row_eq_cur, row_len_cur};
                    SP1HIP_TRY(tabs[cur + 1].alloc((size_t)std::max<uint32_t>((n_out + 1) / 2, 1) * 16, s));   // (written by the round after next)
                    SP1HIP_TRY(tabs[nxt + 1].alloc((size_t)std::max<uint32_t>(n_out, 1) * 16, s));
                    j_materialised = true;
                    hipLaunchKernelGGL(jg_fold_sum_fj, dim3(nb), dim3(256), 0, s, (const Ext*)tabs[cur].p, Jp, n_live, alpha, n_out,
                                       (Ext*)tabs[nxt].p, (Ext*)tabs[nxt + 1].p, sc.tail());
                } else {
                    hipLaunchKernelGGL(jg_fold_sum, dim3(nb), dim3(256), 0, s, (const Ext*)tabs[cur].p, (const Ext*)tabs[cur + 1].p, n_live,
                                       alpha, n_out, (Ext*)tabs[nxt].p, (Ext*)tabs[nxt + 1].p, sc.tail());
                }
            }
            n_live = n_out;
            cur = nxt;
        }
        SP1HIP_LAUNCH_CHECK();
        Ext y0, h4;
        SP1HIP_TRY(sc.finish(nb, &y0, &h4));
        poly = interpolate(y0, h4, claim - y0);
        sumcheck.polys.push_back(poly);
        alpha = observe_and_sample(ch, poly);
        alphas.push_back(alpha);
        claim = poly_eval(poly, alpha);
    }
    sumcheck.point.assign(alphas.rbegin(), alphas.rend());
    sumcheck.eval = claim;
    // component evaluations: fold the last (<= 2 entries) tables on the host
    if (log_m == 0) {
        SP1HIP_REQUIRE(false, "degenerate dense area");
    } else if (log_m == 1) {
        // tables were never materialised: T == 2, fold straight from the base data (cannot happen with lsh >= 1 and
        // two-padding-table rounds unless the area is exactly 2; handled for completeness)
        uint32_t hq[2];
        SP1HIP_TRY(sc.mb.fetch(rounds[0]->d_dense, 2, hq));
        q_eval = kb::ext_from_base(hq[0]) + kb::ext_mul_base(alpha, kb::sub(hq[1], hq[0]));
    } else {
        Ext hq[2] = {kb::ext_zero(), kb::ext_zero()}, hj[2] = {kb::ext_zero(), kb::ext_zero()};
        SP1HIP_TRY(sc.mb.fetch(tabs[cur].p, (size_t)n_live * 4, hq));
        SP1HIP_TRY(sc.mb.fetch(tabs[cur + 1].p, (size_t)n_live * 4, hj));
        q_eval = hq[0] + alpha * (hq[1] - hq[0]);
        j_eval = hj[0] + alpha * (hj[1] - hj[0]);
    }
    (void)j_eval;
    const std::vector<Ext>& final_point = sumcheck.point;

    // ---- jagged-eval proof
    jg_t[2] = std::chrono::steady_clock::now();
    marks.next("jagged_eval");
    Sumcheck jagged_eval;
    SP1HIP_TRY(jagged_eval_prove(prefix, log_m, z_row, z_col, final_point, ch, sc, &jagged_eval));
    jg_t[3] = std::chrono::steady_clock::now();
    marks.next("stacked_basefold_open");

    // ---- dense PCS: observe the claim, evaluate every stacked column at the stack point, BaseFold-open
    for (int k = 0; k < 4; k++) challenger_observe(ch, q_eval.c[k]);
    const std::vector<Ext> stack_point(final_point.end() - lsh, final_point.end());
    DeviceBuf d_eq, d_evals;
    SP1HIP_TRY(d_eq.alloc(((size_t)16) << lsh, s));
    SP1HIP_TRY(sp1hip_partial_lagrange(reinterpret_cast<const sp1hip_ext_t*>(stack_point.data()), lsh, d_eq.u32(), stream));
    std::vector<std::vector<Ext>> batch_evals(n_rounds);
    std::vector<sp1hip_basefold_data_t*> bf;
    std::vector<Ext> flat_claims;
    SP1HIP_TRY(d_evals.alloc((size_t)(*std::max_element(round_widths.begin(), round_widths.end())) * 16, s));
    for (int r = 0; r < n_rounds; r++) {
        const uint32_t w = round_widths[r];
        batch_evals[r].resize(w);
        ScopedTimer t("jagged_batch_evals", s);
        SP1HIP_TRY(sp1hip_mle_eval_columns(rounds[r]->batches.data(), (int)rounds[r]->batches.size(), lsh, d_eq.u32(), d_evals.u32(), stream));
        SP1HIP_TRY(sc.mb.fetch(d_evals.p, (size_t)w * 4, batch_evals[r].data()));
        flat_claims.insert(flat_claims.end(), batch_evals[r].begin(), batch_evals[r].end());
        bf.push_back(rounds[r]->basefold);
    }
    for (auto& e : flat_claims) for (int k = 0; k < 4; k++) challenger_observe(ch, e.c[k]);
    jg_t[4] = std::chrono::steady_clock::now();
    // the BaseFold proof is the first field of JaggedPcsProof: it is written straight into the caller's buffer, the rest behind it
    size_t bf_len = need;
    SP1HIP_TRY(sp1hip_basefold_prove(reinterpret_cast<const sp1hip_ext_t*>(stack_point.data()), lsh, bf.data(), n_rounds,
                                     reinterpret_cast<const sp1hip_ext_t*>(flat_claims.data()), flat_claims.size(), config, ch,
                                     h_proof, &bf_len, stream));

    // ---- bincode(JaggedPcsProof)
    jg_t[5] = std::chrono::steady_clock::now();
    Bytes w;
    w.p = h_proof;
    w.cap = need;
    w.n = bf_len;
    w.u64(n_rounds);
    for (auto& ev : batch_evals) { w.u64(ev.size()); for (auto& e : ev) w.ext(e); w.u64(1); w.u64(ev.size()); }
    sumcheck.write(w);
    jagged_eval.write(w);
    w.u64(n_rounds);
    for (int r = 0; r < n_rounds; r++) {
        w.u64(rounds[r]->row_counts.size());
        for (size_t t = 0; t < rounds[r]->row_counts.size(); t++) { w.u64(rounds[r]->row_counts[t]); w.u64(rounds[r]->column_counts[t]); }
    }
    w.u64(n_rounds);
    for (int r = 0; r < n_rounds; r++) for (int k = 0; k < 8; k++) w.felt(rounds[r]->commit[k]);
    w.ext(q_eval);
    w.u64(max_log_row_count);
    w.u64(log_m);
    if (w.overflow || w.n != need) {
        set_error("internal error: jagged proof size %zu != expected %zu", w.n, need);
        return SP1HIP_ERROR_RUNTIME;
    }
    *proof_len = w.n;
    challenger_restore(challenger, ch);
    if (jg_timing) {
        jg_t[6] = std::chrono::steady_clock::now();
        const auto ms = [&](int a, int b) { return std::chrono::duration<double, std::milli>(jg_t[b] - jg_t[a]).count(); };
        fprintf(stderr, "[sp1hip jagged] set-up %.3f ms | %d sumcheck rounds %.3f | jagged-eval %.3f | column evaluations %.3f | BaseFold %.3f | proof bytes %.3f\n",
                ms(0, 1), log_m, ms(1, 2), ms(2, 3), ms(3, 4), ms(4, 5), ms(5, 6));
    }
    return SP1HIP_SUCCESS;
}

}  // extern "C"
