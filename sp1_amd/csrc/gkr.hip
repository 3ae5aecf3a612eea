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

namespace sp1hip {      // gkr_host.cpp: the interaction-variable rounds in AVX-512 (tables as coefficient planes)
bool gkr_host_simd_available();
void gkr_host_round_sums(const uint32_t* tab, size_t stride, const uint32_t* eq, size_t eq_stride, size_t real_pairs, uint32_t out[6][4]);
void gkr_host_round_fold(const uint32_t* tab, uint32_t* out, size_t stride, size_t real_pairs, const uint32_t alpha[4]);
}
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

// the same into fine-grained HOST memory: system-scope stores (sc0 sc1: past the write-back L2, which would otherwise keep
// the line until the kernel ends)
__device__ __forceinline__ void st_ext_host(Ext* p, uint32_t i, const Ext& e) {
    uint32_t* w = reinterpret_cast<uint32_t*>(p + i);
#pragma unroll
    for (int k = 0; k < 4; k++) __hip_atomic_store(w + k, e.c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
// Per (chip, interaction): program words in device memory, column-major traces. The words are the host's form of one
// interaction of sp1_amd/air.py's InteractionProgram with everything that does not depend on the row folded in once the
// challenges are known: [is_send, n, head[4], vcol(multiplicity), n x (beta index, vcol(value))] — head = alpha + beta_0 kind +
// sum over the CONSTANT values c_j of beta_(1+j) c_j (a third of a core shard's message words are constants: byte opcodes, the
// 16 of a range check, zero operands), the n remaining values name their beta.
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
__global__ __launch_bounds__(256) void first_layer_kernel(const IntDesc* __restrict__ descs, const Ext* __restrict__ betas) {
    const IntDesc d = descs[blockIdx.y];
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < d.rows; r += gridDim.x * blockDim.x) {
        prog_words_t p = (prog_words_t)(uintptr_t)d.prog;
        const bool is_send = p[0] != 0;
        const uint32_t nv = p[1];
        Ext den{{p[2], p[3], p[4], p[5]}};                 // head (see above)
        p += 6;
        uint32_t m = vcol_apply(p, d, r);
        if (!is_send) m = kb::sub(0u, m);
        for (uint32_t j = 0; j < nv; j++) {
            const uint32_t bi = *p++;
            den = kb::ext_add(den, kb::ext_mul_base(ld_ext(betas, bi), vcol_apply(p, d, r)));
        }
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

// ---- the first layer and the two tree levels below it in ONE pass over the traces (round 6). A lane owns FOUR consecutive
// rows of one interaction: it evaluates their fractions (level L), combines them in pairs (level L - 1) and the pairs again
// (level L - 2) in registers and stores all three levels — level L is never read back to build the tree (20 B per entry) and
// level L - 1 (32 B per entry pair) neither: 44 B per first-layer entry move where first_layer_kernel + two transition
// launches moved 80. Column reads are one 16 B load per lane and term (rows % 4 == 0: every table this library's tracers
// make), 1 KiB per wave instruction. Same field operations in the same order as the separate kernels: bit-identical levels.
struct FirstDesc {
    const uint32_t* prog;      // this interaction's words
    const uint32_t* main;      // column-major [main_w][rows]
    const uint32_t* prep;
    uint32_t rows;
    uint32_t* n0_out;          // level L: base numerators [rows]
    Ext* d0_out;               //          ext denominators [rows]
    Ext* n1_out; Ext* d1_out;  // level L - 1 [ceil(rows / 2)]
    Ext* n2_out; Ext* d2_out;  // level L - 2 [ceil(rows / 4)]
};

// vcol over rows 4 q .. 4 q + 3 (rows >= 4 q + 1; lanes past the table's end hold zeros)
__device__ __forceinline__ void vcol_apply4(prog_words_t& p, const FirstDesc& d, uint32_t q, bool vec, uint32_t (&out)[4]) {
    const uint32_t nt = p[0], c0 = p[1];
    p += 2;
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = c0;
    for (uint32_t t = 0; t < nt; t++, p += 3) {
        const uint32_t* col = (p[0] ? d.main : d.prep) + (size_t)p[1] * d.rows;
        const uint32_t wgt = p[2];
        uint32_t v[4];
        if (vec) {
            const q4_t x = gptr(reinterpret_cast<const q4_t*>(col))[q];
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = 4 * q + j < d.rows ? gptr(col)[4 * q + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) out[j] = kb::add(out[j], wgt == kb::R1 ? v[j] : kb::mul(v[j], wgt));
    }
}

__global__ __launch_bounds__(256) void first_layers_kernel(const FirstDesc* __restrict__ descs, const Ext* __restrict__ betas) {
    const FirstDesc d = descs[blockIdx.y];
    const uint32_t quads = (d.rows + 3) / 4, rows1 = (d.rows + 1) / 2, rows2 = (rows1 + 1) / 2;
    const bool vec = (d.rows & 3u) == 0;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
        prog_words_t p = (prog_words_t)(uintptr_t)d.prog;
        const bool is_send = p[0] != 0;
        const uint32_t nv = p[1];
        const Ext head{{p[2], p[3], p[4], p[5]}};
        p += 6;
        uint32_t m[4], val[4];
        vcol_apply4(p, d, q, vec, m);
        Ext den[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { if (!is_send) m[j] = kb::sub(0u, m[j]); den[j] = head; }
        // den += sum_k beta_k value_k with delayed reduction: the four products of a coefficient ride one 64-bit accumulator
        // (4 (p - 1)^2 < 2^64 - 2^58, kb::monty_reduce_wide) — 4 wide multiply-adds per value and row and one reduction per
        // four values, where ext_mul_base + ext_add is four reductions and four modular additions per value
        uint64_t acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[j][c] = 0;
        uint32_t pending = 0;
        for (uint32_t k = 0; k < nv; k++) {
            const uint32_t bi = *p++;
            vcol_apply4(p, d, q, vec, val);
            const Ext b = ld_ext(betas, bi);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[j][c] += (uint64_t)b.c[c] * val[j];
            if (++pending == 4 || k + 1 == nv) {
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int c = 0; c < 4; c++) { den[j].c[c] = kb::add(den[j].c[c], kb::monty_reduce_wide(acc[j][c])); acc[j][c] = 0; }
                pending = 0;
            }
        }
        const uint32_t live = min(4u, d.rows - 4 * q);     // rows of this quad that exist
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((uint32_t)j < live) {
                const uint32_t rp = level_pos(4 * q + j, d.rows);
                gptr(d.n0_out)[rp] = m[j];
                st_ext(d.d0_out, rp, den[j]);
            }
        // level L - 1: rows 2 q, 2 q + 1 (transition_kernel<true>: a missing partner is the padding fraction (0, 1))
        Ext n1[2], d1[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((uint32_t)(2 * h + 1) < live) {
                n1[h] = kb::ext_add(kb::ext_mul_base(den[2 * h + 1], m[2 * h]), kb::ext_mul_base(den[2 * h], m[2 * h + 1]));
                d1[h] = kb::ext_mul(den[2 * h], den[2 * h + 1]);
            } else { n1[h] = kb::ext_from_base(m[2 * h]); d1[h] = den[2 * h]; }
            if ((uint32_t)(2 * h) < live) {
                const uint32_t io = level_pos(2 * q + h, rows1);
                st_ext(d.n1_out, io, n1[h]); st_ext(d.d1_out, io, d1[h]);
            }
        }
        // level L - 2: row q (transition_kernel<false>)
        const uint32_t io = level_pos(q, rows2);
        if (live > 2) {
            st_ext(d.n2_out, io, kb::ext_add(kb::ext_mul(d1[1], n1[0]), kb::ext_mul(d1[0], n1[1])));
            st_ext(d.d2_out, io, kb::ext_mul(d1[0], d1[1]));
        } else { st_ext(d.n2_out, io, n1[0]); st_ext(d.d2_out, io, d1[0]); }
    }
}

// Two tree levels per launch below that: a lane reads four entries of level l, stores two of level l - 1 and one of l - 2.
struct Trans2Desc {
    const Ext* n_in; const Ext* d_in;
    Ext* n1_out; Ext* d1_out; Ext* n2_out; Ext* d2_out;
    uint32_t rows_in;
};
__global__ __launch_bounds__(256) void transition2_kernel(const Trans2Desc* __restrict__ descs) {
    const Trans2Desc d = descs[blockIdx.y];
    const uint32_t quads = (d.rows_in + 3) / 4, rows1 = (d.rows_in + 1) / 2, rows2 = (rows1 + 1) / 2;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
        const uint32_t live = min(4u, d.rows_in - 4 * q);
        Ext n[4], dd[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((uint32_t)j < live) { const uint32_t ip = level_pos(4 * q + j, d.rows_in); n[j] = ld_ext(d.n_in, ip); dd[j] = ld_ext(d.d_in, ip); }
        Ext n1[2], d1[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((uint32_t)(2 * h + 1) < live) {
                n1[h] = kb::ext_add(kb::ext_mul(dd[2 * h + 1], n[2 * h]), kb::ext_mul(dd[2 * h], n[2 * h + 1]));
                d1[h] = kb::ext_mul(dd[2 * h], dd[2 * h + 1]);
            } else if ((uint32_t)(2 * h) < live) { n1[h] = n[2 * h]; d1[h] = dd[2 * h]; }
            if ((uint32_t)(2 * h) < live) {
                const uint32_t io = level_pos(2 * q + h, rows1);
                st_ext(d.n1_out, io, n1[h]); st_ext(d.d1_out, io, d1[h]);
            }
        }
        const uint32_t io = level_pos(q, rows2);
        if (live > 2) {
            st_ext(d.n2_out, io, kb::ext_add(kb::ext_mul(d1[1], n1[0]), kb::ext_mul(d1[0], n1[1])));
            st_ext(d.d2_out, io, kb::ext_mul(d1[0], d1[1]));
        } else { st_ext(d.n2_out, io, n1[0]); st_ext(d.d2_out, io, d1[0]); }
    }
}

// ================================================================ eq tables of one layer
// out holds, for t = 0 .. v, the partial-Lagrange table of the first t coordinates of `pt` at offset 2^t - 1... i.e.
// table t occupies [2^t - 1, 2^(t+1) - 1). Thread g -> (t, i).
struct PointArg { Ext c[32]; };
// out2 = lambda * out: the sums weigh a row's numerators with lambda T and its denominators with T (gkr_pass), so that
// lambda costs no product of its own
// (eq_layer_tables_kernel below builds both: the row prefix tables T / lambda T, and the partial-Lagrange table of ALL
// coordinates of the interaction point — on the device, so that a layer starts without a host-built table and its upload
// on the critical path; the host builds its own copy of every prefix table for the interaction-variable rounds WHILE the
// device runs the row rounds)
// Both tables of a layer in ONE launch (a launch call is ~4.7 us of host time on the layer's critical path): workgroups
// [0, full_blocks) build the interaction table, the rest the row prefix tables.
__global__ __launch_bounds__(256) void eq_layer_tables_kernel(PointArg pi, int niv, Ext* __restrict__ eq_int, uint32_t full_blocks,
                                                              PointArg pa, int v, Ext lambda, Ext* __restrict__ T, Ext* __restrict__ TL) {
    if (blockIdx.x < full_blocks) {
        const uint32_t i = blockIdx.x * 256u + threadIdx.x;
        if (i >= (1u << niv)) return;
        Ext acc = kb::ext_one();
        for (int j = 0; j < niv; j++) {
            const bool bit = (i >> (niv - 1 - j)) & 1u;
            acc = kb::ext_mul(acc, bit ? pi.c[j] : kb::ext_sub(kb::ext_one(), pi.c[j]));
        }
        st_ext(eq_int, i, acc);
        return;
    }
    const uint32_t g = (blockIdx.x - full_blocks) * 256u + threadIdx.x + 1;       // 1 .. 2^(v+1) - 1
    if (g >= (2u << v)) return;
    const int t = 31 - __clz(g);
    const uint32_t i = g - (1u << t);
    Ext acc = kb::ext_one();
    for (int j = 0; j < t; j++) {
        const bool bit = (i >> (t - 1 - j)) & 1u;
        acc = kb::ext_mul(acc, bit ? pa.c[j] : kb::ext_sub(kb::ext_one(), pa.c[j]));
    }
    st_ext(T, g - 1, acc);
    st_ext(TL, g - 1, kb::ext_mul(acc, lambda));
}

// ================================================================ sumcheck over the row variables: two rounds per pass
// A layer's sumcheck  sum_x eq(pt, x) F(x),  F = lambda (n0 d1 + n1 d0) + d0 d1,  binds the row variables last-first.
// One PASS over the tables serves TWO rounds (the reference's look-ahead, /root/reference/sp1-gpu/crates/sys/lib/logup_gkr/
// lookahead.cu:L44-L141, on a different grid and a different work split). With X the last row variable, Y the one before it
// and q the remaining bits of a row index, the four rows 4q .. 4q+3 of a table are its values at (Y, X) in {0,1}^2, every
// table is bilinear in (X, Y) there and F is multi-quadratic, so
//      G(X, Y) = sum_q eq(pt'', q) F_q(X, Y)            (9 coefficients)
// is known from its values on the grid {0, 1, inf}^2 — "inf" = the leading coefficient, i.e. F evaluated on the DIFFERENCES
// of the rows, so the grid costs additions only. The host then has both rounds in closed form, no interpolation and no
// inversion:    p_a(X) = PA eq(pt_a, X) ((1 - pt_b) G(X, 0) + pt_b G(X, 1)),    p_b(Y) = PA eq(pt_a, alpha_a) eq(pt_b, Y) G(alpha_a, Y).
// The next pass folds with BOTH challenges while it loads (rows 4r .. 4r+3 -> row r; the half-folded layer is never
// written) and accumulates the grid of the following two rounds from the folded rows in registers. A layer of v row
// variables is 1 + ceil(v / 2) launches and hand-overs instead of v + 1, and every table is read once per TWO rounds.
//
// Work split: one LANE per OUTPUT ROW — the fold is embarrassingly parallel, every load is a coalesced 16 B / lane run in
// the existing lane-blocked layouts (a lane reads 4 input rows = what one lane of the one-round kernels read), and the
// register footprint is one row, not one quad. The four rows of a grid cell then sit in the four lanes of a DPP quad:
//   * the pure points (0,0), (1,0), (0,1), (1,1) are the rows themselves: lane j evaluates F on its own row;
//   * a difference point needs F on a difference of two rows = B_D d0_D + A_D d1_D: two products, split over the two lanes of
//     the pair (one takes B d0, the other A d1; the sign of a difference cancels in the product), operands by quad_perm;
//   * (inf, inf) is the difference of the differences.
// 7 extension products per row (2 to weigh the row, 2 + 1 + 1 + 1 for the grid) against 6.5 for a lane-per-quad split, and
// lane j's accumulators mean different grid points for different j: they are separated by class (lane mod 4) once, at the
// end of the kernel, into the 10 sums the host wants.
// lambda rides on the eq weight: with w = eq weight of the cell and wl = lambda w (a second table, built with the first),
// F w = (wl n1) d0 + (wl n0 + w d0) d1 is three products per row to form B = wl n1 and A = wl n0 + w d0 and then two per
// grid point instead of four (and on the top layer, whose numerators are base-field words, wl n is four base products).
// Padding rows are the constants (0, 1): F = 1 on every padding cell and the cells' eq mass is 1 - (mass of the real cells),
// as in the one-round kernels (`eq_correction_term`, logup_poly.rs:L521-L530).
struct PassDesc {
    const void* src[4];        // FIRST: {level N, level D, -, -} (row r = entries 2r, 2r+1); else {n0, d0, n1, d1}
    Ext* dst[4];               // folded n0, d0, n1, d1
    uint32_t rows_in;          // real rows of the layer before this pass folds (FIRST: ceil(rows_x / 2))
    uint32_t rows_x;           // FIRST only: real entries of the level
    uint32_t eq_int_index;     // global interaction index
    uint32_t tile0;            // tiled launch: first workgroup of this interaction; FLAT: first lane slot
};

// workgroup -> (interaction, range of its rows): the interactions differ by orders of magnitude in height, so the
// launch is a flat list of equally sized tiles; descs[].tile0 is ascending
__device__ __forceinline__ uint32_t find_desc(const PassDesc* __restrict__ descs, uint32_t K, uint32_t b) {
    uint32_t lo = 0, hi = K;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (descs[mid].tile0 <= b) lo = mid; else hi = mid;
    }
    return lo;
}
struct Row { Ext n0, d0, n1, d1; };

// Layout of the folded tables in scratch: a lane of the next pass wants rows 4k .. 4k+3 of each table, i.e. 64 B at a
// 64 B lane stride — every load instruction of a wave would touch 32 cache lines for 1 KiB (measured: the L1/TA rate
// co-bounds the large rounds with the VALUs; with coalesced addresses the same kernel runs 25 % faster). So inside every
// complete block of 256 rows, row r lives at (r mod 4) * 64 + (r div 4) mod 64: the four loads of a wave are four
// contiguous 1 KiB runs, and the stores of the producing pass (one row per lane) are four 256 B runs each. The
// last (incomplete) block of a table keeps the natural order, so tables never grow and short tables (what the host
// fetches after the last fold) are untouched.
__device__ __forceinline__ uint32_t folded_pos(uint32_t r, uint32_t len) {
    return (r | 255u) < len ? ((r & ~255u) | ((r & 3u) << 6) | ((r >> 2) & 63u)) : r;
}

__device__ __forceinline__ Row padding_row() { return Row{kb::ext_zero(), kb::ext_one(), kb::ext_zero(), kb::ext_one()}; }

// row r of the layer as this pass finds it (before its fold)
template <bool FIRST, bool NBASE>
__device__ __forceinline__ Row load_row(const PassDesc& d, uint32_t r) {
    Row q = padding_row();
    if (r >= d.rows_in) return q;
    if (FIRST) {
        const uint32_t ea = level_pos(2 * r, d.rows_x), eb = level_pos(2 * r + 1, d.rows_x);
        q.n0 = load_n<NBASE>(d.src[0], ea);
        q.d0 = ld_ext((const Ext*)d.src[1], ea);
        if (2 * r + 1 < d.rows_x) { q.n1 = load_n<NBASE>(d.src[0], eb); q.d1 = ld_ext((const Ext*)d.src[1], eb); }
    } else {
        const uint32_t rp = folded_pos(r, d.rows_in);
        q.n0 = ld_ext((const Ext*)d.src[0], rp); q.d0 = ld_ext((const Ext*)d.src[1], rp);
        q.n1 = ld_ext((const Ext*)d.src[2], rp); q.d1 = ld_ext((const Ext*)d.src[3], rp);
    }
    return q;
}

__device__ __forceinline__ Ext lerp(const Ext& a, const Ext& b, const Ext& t) { return kb::ext_add(a, kb::ext_mul(kb::ext_sub(b, a), t)); }   // t: the round's challenge (wave-uniform, second)
__device__ __forceinline__ Row lerp_row(const Row& a, const Row& b, const Ext& t) {
    return Row{lerp(a.n0, b.n0, t), lerp(a.d0, b.d0, t), lerp(a.n1, b.n1, t), lerp(a.d1, b.d1, t)};
}
// the same when the numerators are base-field words (coordinate 0 only): a + t (b - a) is four base products
__device__ __forceinline__ Ext lerp_base(uint32_t a, uint32_t b, const Ext& t) {
    Ext e = kb::ext_mul_base(t, kb::sub(b, a));
    e.c[0] = kb::add(e.c[0], a);
    return e;
}
template <bool NBASE>
__device__ __forceinline__ Row lerp_row_in(const Row& a, const Row& b, const Ext& t) {
    if (!NBASE) return lerp_row(a, b, t);
    return Row{lerp_base(a.n0.c[0], b.n0.c[0], t), lerp(a.d0, b.d0, t), lerp_base(a.n1.c[0], b.n1.c[0], t), lerp(a.d1, b.d1, t)};
}

// output row `ro` of a pass that binds FV variables: rows ro 2^FV .. of the input folded with a0 (last variable), then a1
template <int FV, bool FIRST, bool NBASE>
__device__ __forceinline__ Row fold_row(const PassDesc& d, uint32_t ro, const Ext& a0, const Ext& a1) {
    if constexpr (FV == 0) return load_row<FIRST, NBASE>(d, ro);
    Row in[FV == 0 ? 1 : (1 << FV)];
    // every load of the row is issued before the first use: one exposed memory latency per row
#pragma unroll
    for (int j = 0; j < (1 << FV); j++) in[j] = load_row<FIRST, NBASE>(d, (ro << FV) + j);
    if constexpr (FV == 1) return lerp_row_in<NBASE>(in[0], in[1], a0);
    else {
        const Row lo = lerp_row_in<NBASE>(in[0], in[1], a0), hi = lerp_row_in<NBASE>(in[2], in[3], a0);
        return lerp_row(lo, hi, a1);
    }
}

template <int CTRL>
__device__ __forceinline__ Ext dpp_ext(const Ext& e) {
    return Ext{{p2::dpp_mov<CTRL>(e.c[0]), p2::dpp_mov<CTRL>(e.c[1]), p2::dpp_mov<CTRL>(e.c[2]), p2::dpp_mov<CTRL>(e.c[3])}};
}
__device__ __forceinline__ Ext sel_ext(bool c, const Ext& a, const Ext& b) {
    return Ext{{c ? a.c[0] : b.c[0], c ? a.c[1] : b.c[1], c ? a.c[2] : b.c[2], c ? a.c[3] : b.c[3]}};
}

// per-lane accumulators of a pass that sums SV rounds: what each one means depends on the lane's class (lane mod 2^SV)
template <int SV> struct GridAcc { Ext a[SV == 2 ? 5 : 3]; };
template <int SV>
__device__ __forceinline__ void grid_init(GridAcc<SV>& g) {
#pragma unroll
    for (int i = 0; i < (SV == 2 ? 5 : 3); i++) g.a[i] = kb::ext_zero();
}

// row = this lane's (folded) row; w = eq weight of its cell (zero for lanes past the last real cell), wl = lambda w;
// j = lane's position inside the cell. ALL lanes of the cell must be active. NB: the numerators are base-field words.
template <int SV, bool NB>
__device__ __forceinline__ void grid_accumulate(const Row& row, const Ext& w, const Ext& wl, uint32_t j, GridAcc<SV>& g) {
    const Ext A = kb::ext_add(NB ? kb::ext_mul_base(wl, row.n0.c[0]) : kb::ext_mul(row.n0, wl), kb::ext_mul(row.d0, w));
    const Ext B = NB ? kb::ext_mul_base(wl, row.n1.c[0]) : kb::ext_mul(row.n1, wl);
    g.a[0] = kb::ext_add(g.a[0], kb::ext_add(kb::ext_mul(B, row.d0), kb::ext_mul(A, row.d1)));       // the pure point of this lane
    // difference over the last variable (lanes j ^ 1): the even lane takes B d0, the odd lane A d1 — so every lane KEEPS one
    // pair of operands and SENDS the other pair, the one its partner works on
    const bool lo1 = (j & 1u) == 0;
    const Ext P = sel_ext(lo1, B, A), Q = sel_ext(lo1, row.d0, row.d1), PS = sel_ext(lo1, A, B), QS = sel_ext(lo1, row.d1, row.d0);
    const Ext DP = kb::ext_sub(P, dpp_ext<p2::DPP_QUAD_SWAP1>(PS)), DQ = kb::ext_sub(Q, dpp_ext<p2::DPP_QUAD_SWAP1>(QS));
    g.a[1] = kb::ext_add(g.a[1], kb::ext_mul(DP, DQ));
    if constexpr (SV == 2) {
        // difference over the variable before it (lanes j ^ 2): lanes 0, 1 take B d0, lanes 2, 3 A d1
        const bool lo2 = (j & 2u) == 0;
        const Ext P2 = sel_ext(lo2, B, A), Q2 = sel_ext(lo2, row.d0, row.d1), PS2 = sel_ext(lo2, A, B), QS2 = sel_ext(lo2, row.d1, row.d0);
        const Ext EP = kb::ext_sub(P2, dpp_ext<p2::DPP_QUAD_SWAP2>(PS2)), EQ = kb::ext_sub(Q2, dpp_ext<p2::DPP_QUAD_SWAP2>(QS2));
        g.a[2] = kb::ext_add(g.a[2], kb::ext_mul(EP, EQ));
        // both: lanes j and j ^ 2 hold first differences of the SAME pair of tables (B d0 on lanes 0 / 2, A d1 on 1 / 3), so
        // their difference is the mixed one; computed twice, counted once below
        const Ext FP = kb::ext_sub(DP, dpp_ext<p2::DPP_QUAD_SWAP2>(DP)), FQ = kb::ext_sub(DQ, dpp_ext<p2::DPP_QUAD_SWAP2>(DQ));
        g.a[3] = kb::ext_add(g.a[3], kb::ext_mul(FP, FQ));
    }
    if (j == 0) g.a[SV == 2 ? 4 : 2] = kb::ext_add(g.a[SV == 2 ? 4 : 2], w);                          // eq mass of the real cells
}

// SV = 2: G00 G10 G01 G11 | Gi0 Gi1 | G0i G1i | Gii | Seq   (first index: X = last variable; i = "inf")
// SV = 1: G0 G1 | Gi | Seq
template <int SV> struct GridOut { static constexpr int NS = SV == 2 ? 10 : 4; };
template <int SV>
__device__ __forceinline__ void grid_split(const GridAcc<SV>& g, uint32_t j, const Ext& scale, bool use_scale, Ext (&out)[GridOut<SV>::NS]) {
    const Ext z = kb::ext_zero();
    Ext a[SV == 2 ? 5 : 3];
#pragma unroll
    for (int i = 0; i < (SV == 2 ? 5 : 3); i++) a[i] = use_scale ? kb::ext_mul(g.a[i], scale) : g.a[i];
    if constexpr (SV == 2) {
        out[0] = j == 0 ? a[0] : z; out[1] = j == 1 ? a[0] : z; out[2] = j == 2 ? a[0] : z; out[3] = j == 3 ? a[0] : z;
        out[4] = j < 2 ? a[1] : z; out[5] = j < 2 ? z : a[1];
        out[6] = (j & 1u) == 0 ? a[2] : z; out[7] = (j & 1u) == 0 ? z : a[2];
        out[8] = j < 2 ? a[3] : z;
        out[9] = a[4];
    } else {
        out[0] = j == 0 ? a[0] : z; out[1] = j == 1 ? a[0] : z;
        out[2] = a[1];
        out[3] = a[2];
    }
}

// Small passes (FLAT): most of a proof's passes have a handful of rows per interaction — one workgroup per interaction is
// then 730 workgroups of a few busy lanes each, and what the launch costs is the 730 tickets and partial sums of its tail.
// FLAT launches give every LANE one row instead: lane p of the launch looks its descriptor up in a host-built table
// (flat_index[p]; descs[].tile0 = first slot of the interaction, slots per interaction = rows rounded up to whole cells)
// and weighs its own row with its interaction's eq factor.
struct FlatArgs { const uint16_t* index; uint32_t total_slots; };

// One pass: bind FV variables (fold rows ro 2^FV .. with a0, a1 and store row ro; FV = 0: read only), then accumulate the
// grid of the next SV rounds (SV = 0: the layer's last fold, nothing to sum). T = partial-Lagrange table of the row
// variables that remain after those SV rounds, TL = lambda T.
template <int FV, int SV, bool FIRST, bool NBASE, bool FLAT>
__global__ __launch_bounds__(256) void gkr_pass(const PassDesc* __restrict__ descs, const Ext* __restrict__ eq_int,
                                                const Ext* __restrict__ T, const Ext* __restrict__ TL, Ext a0, Ext a1,
                                                uint32_t* __restrict__ partials, RoundSync rs, uint32_t seq, uint32_t K,
                                                uint32_t tile_size, FlatArgs fa, GateArg gate) {
    static_assert(FV + SV > 0 && FV <= 2 && SV <= 2, "a pass folds and / or sums");
    if (FV > 0 && gate.block != nullptr) {                   // enqueued one hand-over early: the challenges arrive through the gate
        __shared__ uint32_t gate_words[8];
        gate_wait_load(gate, gate_words, 8);
        a0 = Ext{{gate_words[0], gate_words[1], gate_words[2], gate_words[3]}};
        a1 = Ext{{gate_words[4], gate_words[5], gate_words[6], gate_words[7]}};
    }
    constexpr int SVS = SV == 0 ? 1 : SV;                    // (types only; SV = 0 never touches the grid)
    GridAcc<SVS> g;
    grid_init<SVS>(g);
    const uint32_t j = threadIdx.x & ((1u << SV) - 1u);
    // one row: fold, store, weigh, accumulate. valid = the lane has a slot; rows past rows_out inside a real cell are padding
    // (a padding row's numerators are zero: also as base words)
    auto do_row = [&](const PassDesc& d, uint32_t ro, bool valid, const Ext& wi, bool use_wi) {
        const uint32_t rows_out = (d.rows_in + (1u << FV) - 1u) >> FV;
        Row row = padding_row();
        if (valid && ro < rows_out) {
            row = fold_row<FV, FIRST, NBASE>(d, ro, a0, a1);
            if (FV > 0) {
                const uint32_t rq = folded_pos(ro, rows_out);
                if (SV == 0 && rs.host_slot != nullptr) {    // the layer's last fold, published from this kernel (see the tail)
                    st_ext_host(d.dst[0], rq, row.n0); st_ext_host(d.dst[1], rq, row.d0); st_ext_host(d.dst[2], rq, row.n1); st_ext_host(d.dst[3], rq, row.d1);
                } else {
                    st_ext(d.dst[0], rq, row.n0); st_ext(d.dst[1], rq, row.d0); st_ext(d.dst[2], rq, row.n1); st_ext(d.dst[3], rq, row.d1);
                }
            }
        }
        if constexpr (SV > 0) {
            const uint32_t cells = (rows_out + (1u << SV) - 1u) >> SV;
            const uint32_t c = ro >> SV;
            Ext w = kb::ext_zero(), wl = kb::ext_zero();
            if (valid && c < cells) {
                w = ld_ext(T, c); wl = ld_ext(TL, c);
                if (use_wi) { w = kb::ext_mul(w, wi); wl = kb::ext_mul(wl, wi); }
            }
            grid_accumulate<SVS, FIRST && NBASE && FV == 0>(row, w, wl, j, g);
        }
    };
    Ext scale = kb::ext_one();
    if (FLAT) {
        const uint32_t p = blockIdx.x * 256 + threadIdx.x;
        const bool valid = p < fa.total_slots;               // slots come in whole cells: a cell's lanes are valid together
        const PassDesc d = descs[valid ? fa.index[p] : 0];
        const Ext wi = SV > 0 ? ld_ext(eq_int, d.eq_int_index) : kb::ext_one();
        do_row(d, p - d.tile0, valid, wi, true);
    } else {
        const PassDesc d = descs[find_desc(descs, K, blockIdx.x)];
        const uint32_t rows_out = (d.rows_in + (1u << FV) - 1u) >> FV;
        const uint32_t slots = ((rows_out + (1u << SV) - 1u) >> SV) << SV;
        const uint32_t k0 = (blockIdx.x - d.tile0) * tile_size, k1 = min(slots, k0 + tile_size);
        // every lane runs every iteration (the grid exchanges rows between the lanes of a cell)
        for (uint32_t base = k0; base < k1; base += 256) do_row(d, base + threadIdx.x, base + threadIdx.x < k1, scale, false);
        if (SV > 0) scale = ld_ext(eq_int, d.eq_int_index);
    }
    if constexpr (SV > 0) {
        Ext out[GridOut<SVS>::NS];
        grid_split<SVS>(g, j, scale, !FLAT, out);
        rs_finish<GridOut<SVS>::NS>(out, partials, blockIdx.x, gridDim.x, rs, seq);
    } else if (rs.host_slot != nullptr) {
        // The layer's last fold stores its rows (one per interaction) STRAIGHT into the mapped host slot — the descriptors'
        // dst pointers point there — and the workgroup that arrives last publishes the sequence number: the host had been
        // waiting for a one-workgroup copy kernel behind this one (45 us for 46 KB of 4-byte uncached stores, once per layer).
        // Same ordering argument as rs_finish: system-scope stores to fine-grained host memory that the wave has been
        // acknowledged (vmcnt) are visible to the host, and the ticket orders the last workgroup's store behind everybody's.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && rs_ticket_is_last_acq_rel(rs.counter, blockIdx.x, gridDim.x)) rs_publish_seq(rs.host_slot, seq);
    }
}

// ================================================================ trace openings at the final point
struct OpenDesc { const uint32_t* cols; uint32_t rows, width, col0, out0; };   // one group of <= OPEN_COLS columns of one chip
constexpr int OPEN_COLS = 4, OPEN_ROWS = 16384;      // (8 columns per workgroup, sharing one read of the eq slice, measured slower: 1.33 vs 0.78 ms)
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
// out[j] = sum over the chunks of partials[chunk][j]: 64 words per workgroup, four chunk lanes per word folded through LDS
// (one lane per word walking all chunks was a chain of n_chunks dependent adds: 50 us for 256 chunks)
__global__ __launch_bounds__(256) void open_sum_kernel(const uint32_t* __restrict__ partials, uint32_t n_chunks, uint32_t n_words, uint32_t* __restrict__ out) {
    __shared__ uint32_t sm[256];
    const uint32_t jl = threadIdx.x & 63u, cl = threadIdx.x >> 6, j = blockIdx.x * 64u + jl;
    uint32_t acc = 0;
    if (j < n_words)
        for (uint32_t c = cl; c < n_chunks; c += 4) acc = kb::add(acc, partials[(size_t)c * n_words + j]);
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (cl == 0 && j < n_words) out[j] = kb::add(kb::add(sm[jl], sm[64 + jl]), kb::add(sm[128 + jl], sm[192 + jl]));
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
    // per global interaction: is_send, kind (canonical), multiplicity and values as (constant flag, words) — words = the vcol in
    // device form [n_terms, constant (Montgomery), n_terms x (is_main, column, weight (Montgomery))]
    struct ParsedInt { uint32_t is_send, kind; std::vector<uint32_t> mult; std::vector<std::vector<uint32_t>> values; };
    std::vector<ParsedInt> progs;
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
            const uint32_t nv = p[2];
            SP1HIP_REQUIRE(p[0] <= 1 && p[1] < kb::P && nv < 4096, "bad interaction header");      // (the Keccak bus carries 106 values per message)
            ParsedInt pi{p[0], p[1], {}, std::vector<std::vector<uint32_t>>(nv)};
            p += 3;
            SP1HIP_REQUIRE(vcol(pi.mult), "bad multiplicity column");
            for (uint32_t j = 0; j < nv; j++) SP1HIP_REQUIRE(vcol(pi.values[j]), "bad value column (index or weight out of range)");
            max_arity = std::max<size_t>(max_arity, nv + 1);
            progs.push_back(std::move(pi));
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
    // The pass descriptors (~10 MB at 730 interactions) go up on the caller's SIDE stream while the first layer and the
    // fraction tree run: on `s` the copy sat between the last tree kernel and the first hand-over (0.2 ms of an idle GPU).
    // Their buffer is taken from the arena HERE, before anything of this call is enqueued, and the side stream waits for
    // this point of `s` — a recycled block's previous user is then out of the way (the arena orders reuse by stream).
    const uint32_t FLAT_MAX_SLOTS = [] { const char* e = getenv("SP1HIP_GKR_FLAT_SLOTS"); return e ? (uint32_t)atoi(e) : 65536u; }();   // read per call (tests)
    size_t n_passes_total = 0;
    for (int v = 1; v <= L - 1; v++) n_passes_total += (size_t)(v + 1) / 2 + 1;     // (0, .), the folds in pairs, the last fold
    hipStream_t side = nullptr;
    hipEvent_t* side_ev = nullptr;
    SP1HIP_TRY(aux_stream_for(s, 2, &side, &side_ev));
    // (their buffers come from the SIDE stream's share of the arena: a recycled block's previous user ran on that stream,
    // so the copies need not wait for anything on `s`; the blocks go back when this call has seen its last result)
    DeviceBuf d_all, d_flat;
    SP1HIP_TRY(d_all.alloc(std::max<size_t>(n_passes_total * K, 1) * sizeof(PassDesc), side));
    SP1HIP_TRY(d_flat.alloc((n_passes_total * (size_t)FLAT_MAX_SLOTS + 4) * sizeof(uint16_t), side));    // a flat pass indexes <= FLAT_MAX_SLOTS slots
    PinnedStage side_stage;
    SP1HIP_TRY(side_stage.init(side));
    // ---- first layer
    PinnedStage stage;                                       // small uploads (round_sync.hpp)
    SP1HIP_TRY(stage.init(s));
    DeviceBuf d_progs, d_betas, d_descs;
    uint32_t max_rows = 0;
    bool fused_tree = false;
    std::vector<uint32_t> flat;                              // upload sources live to the end of the call
    std::vector<IntDesc> descs(K);
    {
        // device programs (see first_layer_kernel): the row-independent part of every denominator is folded into `head` here
        std::vector<size_t> poff(K);
        for (uint32_t i = 0; i < K; i++) {
            poff[i] = flat.size();
            const ParsedInt& pi = progs[i];
            Ext head = alpha + kb::ext_mul_base(betas[0], kb::to_monty(pi.kind));
            uint32_t live = 0;
            for (size_t j = 0; j < pi.values.size(); j++) {
                if (pi.values[j][0] == 0) head = head + kb::ext_mul_base(betas[1 + j], pi.values[j][1]);   // no terms: the constant
                else live++;
            }
            flat.insert(flat.end(), {pi.is_send, live, head.c[0], head.c[1], head.c[2], head.c[3]});
            flat.insert(flat.end(), pi.mult.begin(), pi.mult.end());
            for (size_t j = 0; j < pi.values.size(); j++) {
                if (pi.values[j][0] == 0) continue;
                flat.push_back((uint32_t)(1 + j));
                flat.insert(flat.end(), pi.values[j].begin(), pi.values[j].end());
            }
        }
        SP1HIP_TRY(upload(d_progs, flat.data(), flat.size() * 4, s, stage));
        SP1HIP_TRY(upload(d_betas, betas.data(), betas.size() * 16, s, stage));
        for (uint32_t i = 0; i < K; i++) {
            const ChipInfo& c = info[int_chip[i]];
            descs[i] = IntDesc{d_progs.u32() + poff[i], c.d_main, c.d_prep, c.rows, (uint32_t*)n_ptr(L, i), d_ptr(L, i)};
            max_rows = std::max(max_rows, c.rows);
        }
        // L >= 3: the first layer and the two levels below it come out of ONE pass over the traces (first_layers_kernel), the
        // rest of the tree two levels per launch (transition2_kernel; a last single level by transition_kernel). SP1HIP_GKR_FUSED=0:
        // the separate kernels (first_layer_kernel, one transition launch per level) — same words in every level (tests)
        fused_tree = L >= 3 && [] { const char* e = getenv("SP1HIP_GKR_FUSED"); return !e || atoi(e) != 0; }();
        if (fused_tree) {
            std::vector<FirstDesc> fdescs(K);
            for (uint32_t i = 0; i < K; i++)
                fdescs[i] = FirstDesc{descs[i].prog, descs[i].main, descs[i].prep, descs[i].rows, descs[i].n_out, descs[i].d_out,
                                      (Ext*)n_ptr(L - 1, i), d_ptr(L - 1, i), (Ext*)n_ptr(L - 2, i), d_ptr(L - 2, i)};
            SP1HIP_TRY(upload(d_descs, fdescs.data(), fdescs.size() * sizeof(FirstDesc), s, stage));
            if (max_rows) {
                ScopedTimer t("gkr_first_layer", s);
                hipLaunchKernelGGL(first_layers_kernel, dim3(tiles_for((max_rows + 3) / 4), K), dim3(256), 0, s, (const FirstDesc*)d_descs.p,
                                   (const Ext*)d_betas.p);
                SP1HIP_LAUNCH_CHECK();
            }
        } else {
            SP1HIP_TRY(upload(d_descs, descs.data(), descs.size() * sizeof(IntDesc), s, stage));
            if (max_rows) {
                ScopedTimer t("gkr_first_layer", s);
                hipLaunchKernelGGL(first_layer_kernel, dim3(tiles_for(max_rows), K), dim3(256), 0, s, (const IntDesc*)d_descs.p,
                                   (const Ext*)d_betas.p);
                SP1HIP_LAUNCH_CHECK();
            }
        }
    }
    mark("first layer enqueued");
    // ---- fraction tree
    // every level's descriptors are planned and uploaded once: the tree is built by back-to-back launches
    DeviceBuf d_trans, d_trans2;
    {
        const int top = fused_tree ? L - 2 : L;              // the highest level that still has to be combined downwards
        std::vector<Trans2Desc> t2;                          // launches (top -> top - 2), (top - 2 -> top - 4), ...
        std::vector<TransDesc> t1;                           // then at most one single level (or, unfused, every level)
        std::vector<uint32_t> t2_mr, t1_mr;
        std::vector<int> t1_level;
        int l = top;
        if (fused_tree)
            for (; l >= 3; l -= 2) {
                uint32_t mr = 0;
                for (uint32_t i = 0; i < K; i++) {
                    const uint32_t rin = rows_at(info[int_chip[i]].rows, l);
                    t2.push_back(Trans2Desc{(const Ext*)n_ptr(l, i), d_ptr(l, i), (Ext*)n_ptr(l - 1, i), d_ptr(l - 1, i), (Ext*)n_ptr(l - 2, i), d_ptr(l - 2, i), rin});
                    mr = std::max(mr, (rin + 3) / 4);
                }
                t2_mr.push_back(mr);
            }
        for (; l >= 2; l--) {
            uint32_t mr = 0;
            for (uint32_t i = 0; i < K; i++) {
                const uint32_t rin = rows_at(info[int_chip[i]].rows, l);
                t1.push_back(TransDesc{n_ptr(l, i), d_ptr(l, i), (Ext*)n_ptr(l - 1, i), d_ptr(l - 1, i), rin});
                mr = std::max(mr, (rin + 1) / 2);
            }
            t1_mr.push_back(mr);
            t1_level.push_back(l);
        }
        if (!t2.empty()) SP1HIP_TRY(upload(d_trans2, t2.data(), t2.size() * sizeof(Trans2Desc), s, stage));
        if (!t1.empty()) SP1HIP_TRY(upload(d_trans, t1.data(), t1.size() * sizeof(TransDesc), s, stage));
        for (size_t k = 0; k < t2_mr.size(); k++) {
            if (!t2_mr[k]) continue;
            ScopedTimer t("gkr_transition", s);
            hipLaunchKernelGGL(transition2_kernel, dim3(tiles_for(t2_mr[k]), K), dim3(256), 0, s, (const Trans2Desc*)d_trans2.p + k * K);
            SP1HIP_LAUNCH_CHECK();
        }
        for (size_t k = 0; k < t1_mr.size(); k++) {
            if (!t1_mr[k]) continue;
            const TransDesc* d_t = (const TransDesc*)d_trans.p + k * K;
            ScopedTimer t("gkr_transition", s);
            if (t1_level[k] == L) hipLaunchKernelGGL(transition_kernel<true>, dim3(tiles_for(t1_mr[k]), K), dim3(256), 0, s, d_t);
            else hipLaunchKernelGGL(transition_kernel<false>, dim3(tiles_for(t1_mr[k]), K), dim3(256), 0, s, d_t);
            SP1HIP_LAUNCH_CHECK();
        }
    }
    mark("tree enqueued");
    Mailbox mb;                                              // device -> host hand-overs outside the sumcheck rounds
    SP1HIP_TRY(mb.init(s));
    // (the descriptors of every pass depend on shapes only: they are planned and uploaded WHILE the GPU builds the first
    // layer and the fraction tree — milliseconds of host work at 730 interactions x ~130 passes that would otherwise sit on
    // the critical path behind the first hand-over)
    // ---- GKR rounds, layer v = 1 .. L-1 (reads level v + 1)
    struct RoundOut { Ext n0, n1, d0, d1; std::vector<Poly4> polys; Ext claimed_sum, eval; std::vector<Ext> point; };
    std::vector<RoundOut> rounds;
    DeviceBuf d_eq_int, d_T, d_TL, d_partials, scratch[2];
    SP1HIP_TRY(d_eq_int.alloc((size_t)W * 16, s));
    SP1HIP_TRY(d_T.alloc(((size_t)2 << std::max(L - 1, 1)) * 16, s));
    SP1HIP_TRY(d_TL.alloc(((size_t)2 << std::max(L - 1, 1)) * 16, s));
    // folded tables: 4 vectors per interaction, at most ceil(rows(level v+1) / 4) entries each after the first fold
    size_t scratch_entries = 0;
    for (uint32_t i = 0; i < K; i++) scratch_entries += (rows_at(info[int_chip[i]].rows, L) + 3) / 4 + 1;
    for (int b = 0; b < 2; b++) SP1HIP_TRY(scratch[b].alloc(std::max<size_t>(scratch_entries, 1) * 64, s));
    auto scratch_ptr = [&](int b, uint32_t i, int which, const std::vector<size_t>& so) -> Ext* { return scratch[b].ext() + 4 * so[i] + (size_t)which * (so[i + 1] - so[i]); };
    // The passes of a layer with v row variables (two rounds per pass, gkr_pass): (fold 0, sum s0 = min(2, v)), then
    // (fold what was just summed, sum min(2, what remains)) until nothing remains; the last pass only folds.
    struct PassShape { int fv, sv; bool first; uint32_t tiles, tile_size, total_slots; size_t flat_off; bool flat; };
    std::vector<PassShape> shapes;
    std::vector<uint16_t> flat_index;                        // FLAT launches: slot -> descriptor, all launches back to back
    // workgroups of a large pass. While every workgroup paid an L2 write-back and a serialised ticket in its tail
    // (round_sync.hpp) one resident set — 256 CUs x 4 workgroups — was the optimum; without them finer tiles balance the
    // tail better (sweep of round 2 on the one-round kernels: 4096 best, flat between 2048 and 8192). Round 4, two-round passes on
    // the real-chip shard (alternating runs on three boxes): 1024-3072 tiles are within 0.3 ms of each other, 4096 is 0.7-1.0 ms
    // slower over the stage, 8192 / 16384 1.4 / 3.2 ms slower — a pass carries more state per workgroup than a round kernel did.
    static const uint32_t TARGET_TILES = [] { const char* e = getenv("SP1HIP_GKR_TILES"); return e ? std::max<uint32_t>((uint32_t)atoi(e), 1u) : 2048u; }();
    std::vector<PassDesc> all_descs;
    all_descs.reserve(n_passes_total * K);                  // ~10 MB at 730 interactions: no regrowth copies while planning
    // a layer's last fold (one row per interaction) goes straight to the mailbox slot, 16-byte aligned behind word 0
    const bool direct_final = (size_t)K * 16 + 4 <= MAILBOX_WORDS;
    Ext* const host_rows = reinterpret_cast<Ext*>(mb.h_slot + 4);
    for (int v = 1; v <= L - 1; v++) {
        std::vector<uint32_t> rows_in(K);
        for (uint32_t i = 0; i < K; i++) rows_in[i] = (rows_at(info[int_chip[i]].rows, v + 1) + 1) / 2;
        int cur = 0, t = v, fv = 0, pass = 0;
        std::vector<size_t> so_prev, so_next;
        for (;;) {
            const int sv = std::min(2, t - fv);              // variables left after this pass's fold: t - fv
            const bool first = pass <= 1;                    // pass 0 only sums, so pass 1 still reads the level
            if (fv > 0) { so_next.assign(K + 1, 0); for (uint32_t i = 0; i < K; i++) so_next[i + 1] = so_next[i] + ((rows_in[i] + (1u << fv) - 1) >> fv); }
            uint64_t total_slots = 0;
            auto slots_of = [&](uint32_t rin) -> uint32_t { const uint32_t ro = (rin + (1u << fv) - 1) >> fv; return ((ro + (1u << sv) - 1) >> sv) << sv; };
            for (uint32_t i = 0; i < K; i++) total_slots += slots_of(rows_in[i]);
            const bool flat = total_slots <= FLAT_MAX_SLOTS && K <= 65536;
            const uint32_t tile_size = flat ? 1u : (uint32_t)std::max<uint64_t>(256, ((total_slots + TARGET_TILES - 1) / TARGET_TILES + 255) / 256 * 256);
            const size_t flat_off = flat_index.size();
            all_descs.resize(all_descs.size() + K);
            PassDesc* out = all_descs.data() + all_descs.size() - K;
            uint32_t tile0 = 0;
            for (uint32_t i = 0; i < K; i++) {
                PassDesc& d = out[i];
                d = PassDesc{};
                d.rows_in = rows_in[i]; d.eq_int_index = i;
                if (first) { d.src[0] = n_ptr(v + 1, i); d.src[1] = d_ptr(v + 1, i); d.rows_x = rows_at(info[int_chip[i]].rows, v + 1); }
                else for (int w = 0; w < 4; w++) d.src[w] = scratch_ptr(cur ^ 1, i, w, so_prev);
                if (fv > 0) for (int w = 0; w < 4; w++)
                    d.dst[w] = (sv == 0 && direct_final) ? host_rows + 4 * so_next[i] + (size_t)w * (so_next[i + 1] - so_next[i]) : scratch_ptr(cur, i, w, so_next);
                d.tile0 = tile0;
                const uint32_t n_tiles = (slots_of(rows_in[i]) + tile_size - 1) / tile_size;
                if (flat) flat_index.insert(flat_index.end(), n_tiles, (uint16_t)i);
                tile0 += n_tiles;
            }
            if (flat) shapes.push_back(PassShape{fv, sv, first, std::max<uint32_t>((tile0 + 255) / 256, 1), tile_size, tile0, flat_off, true});
            else shapes.push_back(PassShape{fv, sv, first, std::max<uint32_t>(tile0, 1), tile_size, tile0, 0, false});
            if (fv > 0) { for (uint32_t i = 0; i < K; i++) rows_in[i] = (rows_in[i] + (1u << fv) - 1) >> fv; so_prev = so_next; cur ^= 1; }
            t -= fv;
            pass++;
            if (sv == 0) break;
            fv = sv;
        }
    }
    {   // partial sums: one slot per workgroup of the largest launch, 10 sums each
        uint32_t max_tiles = 1;
        for (auto& sh : shapes) max_tiles = std::max(max_tiles, sh.tiles);
        SP1HIP_TRY(d_partials.alloc((size_t)max_tiles * 160, s));
    }
    mark("pass descriptors planned");
    SP1HIP_REQUIRE(all_descs.size() == n_passes_total * K, "internal error: pass count");
    SP1HIP_TRY(side_stage.upload(d_all.p, all_descs.data(), all_descs.size() * sizeof(PassDesc)));
    if (flat_index.empty()) flat_index.push_back(0);
    flat_index.resize((flat_index.size() + 1) / 2 * 2);
    SP1HIP_REQUIRE(flat_index.size() * sizeof(uint16_t) <= d_flat.n, "internal error: flat index size");
    SP1HIP_TRY(side_stage.upload(d_flat.p, flat_index.data(), flat_index.size() * sizeof(uint16_t)));
    SP1HIP_HIP(hipEventRecord(side_ev[1], side));
    SP1HIP_HIP(hipStreamWaitEvent(s, side_ev[1], 0));        // (enqueued behind the tree: the copies have long finished by then)
    mark("pass descriptors uploaded");
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

    size_t launch_idx = 0;                                   // next pass: shapes[launch_idx], K descriptors of d_all
    RoundSyncHost rsync;
    SP1HIP_TRY(rsync.init(s));
    // SP1HIP_GATE=1 (off by default): small passes are enqueued one hand-over early and wait for their challenges at a HostGate
    // (round_sync.hpp), so that the dispatch of a pass overlaps the host's half of the round trip. Measured on the fibonacci shard
    // (eight alternating runs): 25.3-25.6 ms with the gate against 24.5-25.1 without — the two reads of mapped host memory a gated
    // workgroup starts with (ticket, then challenges: ~2 us each across PCIe) cost what the overlapped dispatch saves. Kept as an
    // A/B knob; the proof bytes are the same either way (tests/test_gpu_gkr.py). SP1HIP_GATE_MAX_TILES: what "small" means.
    const bool gate_on = [] { const char* e = getenv("SP1HIP_GATE"); return e && e[0] == '1'; }();
    const uint32_t gate_max_tiles = [] { const char* e = getenv("SP1HIP_GATE_MAX_TILES"); return e ? (uint32_t)strtoul(e, nullptr, 10) : 256u; }();
    HostGate gate;                                           // (declared after rsync / mb: destroyed first, so an early exit opens it before they drain)
    if (gate_on) SP1HIP_TRY(gate.init(s));
    const Ext one = kb::ext_one(), zero = kb::ext_zero(), inv8 = kb::ext_inv(ext_c(8)), inv2 = kb::ext_inv(ext_c(2)), four = ext_c(4);
    uint32_t h_sums[40];
    // the cubic  scale (1 - pt + (2 pt - 1) X) (q0 + q1 X + q2 X^2)
    auto eq_times_quadratic = [&](const Ext& scale, const Ext& pt, const Ext (&q)[3]) -> Poly4 {
        const Ext e0 = scale * (one - pt), e1 = scale * (pt + pt - one);
        return Poly4{e0 * q[0], e0 * q[1] + e1 * q[0], e0 * q[2] + e1 * q[1], e1 * q[2]};
    };
    // the quadratic with values g0, g1 at 0, 1 and leading coefficient gi
    auto quadratic = [&](const Ext& g0, const Ext& g1, const Ext& gi, Ext (&q)[3]) { q[0] = g0; q[1] = g1 - g0 - gi; q[2] = gi; };
    auto eval_quadratic = [&](const Ext (&q)[3], const Ext& x) -> Ext { return (q[2] * x + q[1]) * x + q[0]; };

    const bool simd = gkr_host_simd_available();             // AVX-512 host rounds: one thread, the helpers stay asleep
    HostPar::Scope par;                                      // helper threads for the host loops between hand-overs
    par.park();                                              // asleep through the device rounds; woken ahead of each layer's host rounds
    double dbg_rows = 0, dbg_int = 0, dbg_head = 0;
    auto dbg_t = std::chrono::steady_clock::now();
    auto dbg_pass_t = dbg_t;
    double dbg_wait = 0, dbg_host = 0, dbg_launch = 0;
    int dbg_passes = 0;
    bool dbg_pass_pending = false;
    for (int v = 1; v <= L - 1; v++) {
        if (gkr_debug) dbg_t = std::chrono::steady_clock::now();
        const Ext lambda = challenger_sample_ext(ch);
        RoundOut ro;
        Ext claim = num_eval * lambda + den_eval;
        ro.claimed_sum = claim;
        const std::vector<Ext> int_point(eval_point.begin(), eval_point.begin() + niv), row_point(eval_point.begin() + niv, eval_point.end());
        // the device builds its own tables: eq over the interaction point, and over every prefix of the row point (x 1 and x lambda)
        SP1HIP_REQUIRE(niv <= 32 && v <= 32, "point too long");
        {
            PointArg pi{}, pa{};
            for (int j = 0; j < niv; j++) pi.c[j] = int_point[j];
            for (int j = 0; j < v; j++) pa.c[j] = row_point[j];
            const uint32_t full_blocks = (W + 255) / 256, prefix_blocks = ((2u << v) + 255) / 256;
            hipLaunchKernelGGL(eq_layer_tables_kernel, dim3(full_blocks + prefix_blocks), dim3(256), 0, s, pi, niv, d_eq_int.ext(), full_blocks, pa, v,
                               lambda, d_T.ext(), d_TL.ext());
            SP1HIP_LAUNCH_CHECK();
        }
        // Lagrange tables of every prefix of the interaction point (eq_tabs[m]: the first m coordinates, 2^m entries) for the
        // HOST rounds at the end of the layer: built lazily, after the first pass of the layer has been launched
        std::vector<std::vector<Ext>> eq_tabs(niv + 1);
        std::vector<std::vector<uint32_t>> eq_soa(niv + 1);
        auto eq_plane_stride = [](int m) { return ((((size_t)1 << m) + 15) / 16) * 16 + 16; };
        bool eq_tabs_built = false;
        auto build_eq_tabs = [&]() {
            if (eq_tabs_built) return;
            eq_tabs_built = true;
            eq_tabs[0] = {one};
            for (int m = 0; m < niv; m++) {
                const std::vector<Ext>& ev = eq_tabs[m];
                std::vector<Ext>& nx = eq_tabs[m + 1];
                nx.resize(ev.size() * 2);
                const Ext x = int_point[m];
                for (size_t i = 0; i < ev.size(); i++) { const Ext pr = ev[i] * x; nx[2 * i] = ev[i] - pr; nx[2 * i + 1] = pr; }
            }
            if (simd)                                         // the same tables as coefficient planes (gkr_host.cpp)
                for (int m = 1; m <= niv; m++) {
                    const size_t es = eq_plane_stride(m);
                    eq_soa[m].assign(4 * es, 0u);
                    for (size_t i = 0; i < eq_tabs[m].size(); i++)
                        for (int k = 0; k < 4; k++) eq_soa[m][(size_t)k * es + i] = eq_tabs[m][i].c[k];
                }
        };
        auto T_of = [&](int t) -> const Ext* { return d_T.ext() + (((size_t)1 << t) - 1); };
        auto TL_of = [&](int t) -> const Ext* { return d_TL.ext() + (((size_t)1 << t) - 1); };

        std::vector<Ext> alphas;
        Ext PA = one;                                        // eq factor of the row variables bound so far
        Ext a0 = zero, a1 = zero;                            // the challenges the next pass folds with
        int t = v;                                           // row variables not yet bound by a FOLD
        size_t final_rows_total = 0;
        // one pass of the layer: shapes[idx], t_after = row variables left once it has folded. gate_slot: the pass is enqueued behind
        // a HostGate and reads its challenges there; else they are the arguments. *seq_out = the number of its hand-over (sv > 0).
        auto launch_pass = [&](size_t idx, int t_after, GateArg gate_arg, const Ext& c0, const Ext& c1, uint32_t* seq_out) -> int {
            const PassShape shape = shapes[idx];
            const PassDesc* d_descs = (const PassDesc*)d_all.p + idx * K;
            const int fv = shape.fv, sv = shape.sv;
            const FlatArgs fa{(const uint16_t*)d_flat.p + shape.flat_off, shape.total_slots};
            const bool nbase = shape.first && v + 1 == L;
            const bool publish_rows = sv == 0 && direct_final;
            if (publish_rows) { rsync.pending = true; mb.pending = true; }       // (an early error return must drain the stream first)
            const RoundSync rs = sv > 0 ? rsync.next() : publish_rows ? RoundSync{rsync.d_counter, (volatile uint32_t*)mb.h_slot} : RoundSync{};
            *seq_out = rsync.seq;
            const Ext* Tp = sv > 0 ? T_of(t_after - sv) : (const Ext*)nullptr;
            const Ext* TLp = sv > 0 ? TL_of(t_after - sv) : (const Ext*)nullptr;
            ScopedTimer tm(fv == 0 ? "gkr_pass_sum" : sv == 0 ? "gkr_pass_fold" : "gkr_pass_fold_sum", s);
#define SP1HIP_GKR_PASS(FV, SV, F, NB, FL) hipLaunchKernelGGL((gkr_pass<FV, SV, F, NB, FL>), dim3(shape.tiles), dim3(256), 0, s, d_descs, (const Ext*)d_eq_int.p, Tp, TLp, c0, c1, d_partials.u32(), rs, publish_rows ? mb.seq + 1 : rsync.seq, K, shape.tile_size, fa, gate_arg)
#define SP1HIP_GKR_PASS_FL(FV, SV, F, NB) do { if (shape.flat) SP1HIP_GKR_PASS(FV, SV, F, NB, true); else SP1HIP_GKR_PASS(FV, SV, F, NB, false); } while (0)
#define SP1HIP_GKR_PASS_SRC(FV, SV) do { if (nbase) SP1HIP_GKR_PASS_FL(FV, SV, true, true); else if (shape.first) SP1HIP_GKR_PASS_FL(FV, SV, true, false); else SP1HIP_GKR_PASS_FL(FV, SV, false, false); } while (0)
            if (fv == 0 && sv == 2) { if (nbase) SP1HIP_GKR_PASS_FL(0, 2, true, true); else SP1HIP_GKR_PASS_FL(0, 2, true, false); }
            else if (fv == 0 && sv == 1) { if (nbase) SP1HIP_GKR_PASS_FL(0, 1, true, true); else SP1HIP_GKR_PASS_FL(0, 1, true, false); }
            else if (fv == 2 && sv == 2) SP1HIP_GKR_PASS_SRC(2, 2);
            else if (fv == 2 && sv == 1) SP1HIP_GKR_PASS_SRC(2, 1);
            else if (fv == 2 && sv == 0) SP1HIP_GKR_PASS_SRC(2, 0);
            else if (fv == 1 && sv == 0) SP1HIP_GKR_PASS_SRC(1, 0);
            else { set_error("internal error: GKR pass shape (%d, %d)", fv, sv); return SP1HIP_ERROR_RUNTIME; }
#undef SP1HIP_GKR_PASS_SRC
#undef SP1HIP_GKR_PASS_FL
#undef SP1HIP_GKR_PASS
            SP1HIP_LAUNCH_CHECK();
            return SP1HIP_SUCCESS;
        };
        bool pre = false;                                    // the pass about to be handled is already enqueued (behind the gate, opened)
        uint32_t pre_ticket = 0, pre_seq = 0;
        for (;;) {                                           // the passes of the layer (two rounds each)
            const PassShape shape = shapes[launch_idx];
            const size_t idx = launch_idx++;
            const int fv = shape.fv, sv = shape.sv;
            t -= fv;                                         // variables left once this pass has folded
            if (!simd && t == sv) par.wake();                // the helpers' wake-up hides behind the layer's last two passes
            const auto dbg_l0 = std::chrono::steady_clock::now();
            if (gkr_debug && dbg_pass_pending) dbg_host += std::chrono::duration<double, std::milli>(dbg_l0 - dbg_pass_t).count();
            dbg_pass_pending = false;
            uint32_t my_seq = pre_seq;
            if (!pre) SP1HIP_TRY(launch_pass(idx, t, GateArg{nullptr, 0u}, a0, a1, &my_seq));
            pre = false;
            if (gkr_debug) dbg_launch += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_l0).count();
            build_eq_tabs();                                 // (first pass of the layer only) host work behind a running kernel
            if (sv == 0) break;
            // the NEXT pass, one hand-over early when it is small (its duration is latency: the launch would be on the critical path)
            if (gate_on && shapes[launch_idx].tiles <= gate_max_tiles) {
                const GateArg ga = gate.arm();
                pre_ticket = ga.ticket;
                pre = true;                                  // (armed: from here on the ticket must be opened, whatever happens)
                SP1HIP_TRY(launch_pass(launch_idx, t - shapes[launch_idx].fv, ga, zero, zero, &pre_seq));
            }
            const int ns = sv == 2 ? 10 : 4;
            const auto dbg_w0 = std::chrono::steady_clock::now();
            SP1HIP_TRY(rsync.wait_for(my_seq, h_sums, 4 * ns));
            if (gkr_debug) { const auto now = std::chrono::steady_clock::now(); dbg_wait += std::chrono::duration<double, std::milli>(now - dbg_w0).count(); dbg_pass_t = now; dbg_passes++; dbg_pass_pending = true; }
            Ext S[10];
            memcpy(S, h_sums, 16 * (size_t)ns);
            const Ext pt_a = row_point[t - 1];               // the last unbound variable
            if (sv == 2) {
                // S: G00 G10 G01 G11 | Gi0 Gi1 | G0i G1i | Gii | Seq (first index: the last variable). The padding cells carry
                // F = 1, i.e. their eq mass on the constant coefficient: on the four finite grid points
                const Ext pt_b = row_point[t - 2];
                const Ext corr = one - S[9];
                Ext c0[3], c1[3], ci[3];
                quadratic(S[0] + corr, S[1] + corr, S[4], c0);             // G(X, 0)
                quadratic(S[2] + corr, S[3] + corr, S[5], c1);             // G(X, 1)
                quadratic(S[6], S[7], S[8], ci);                           // leading coefficient in Y
                Ext h[3];
                for (int k = 0; k < 3; k++) h[k] = c0[k] + pt_b * (c1[k] - c0[k]);
                Poly4 poly = eq_times_quadratic(PA, pt_a, h);
                ro.polys.push_back(poly);
                for (auto& c : poly) observe_ext(ch, c);
                a0 = challenger_sample_ext(ch);
                alphas.push_back(a0);
                claim = poly_eval(poly, a0);
                PA = PA * (pt_a * a0 + (one - pt_a) * (one - a0));
                const Ext g0 = eval_quadratic(c0, a0), g1 = eval_quadratic(c1, a0), gi = eval_quadratic(ci, a0);
                Ext gy[3];
                quadratic(g0, g1, gi, gy);                                 // G(a0, Y)
                poly = eq_times_quadratic(PA, pt_b, gy);
                ro.polys.push_back(poly);
                for (auto& c : poly) observe_ext(ch, c);
                a1 = challenger_sample_ext(ch);
                alphas.push_back(a1);
                claim = poly_eval(poly, a1);
                PA = PA * (pt_b * a1 + (one - pt_b) * (one - a1));
            } else {
                // S: G0 G1 | Gi | Seq
                const Ext corr = one - S[3];
                Ext h[3];
                quadratic(S[0] + corr, S[1] + corr, S[2], h);
                Poly4 poly = eq_times_quadratic(PA, pt_a, h);
                ro.polys.push_back(poly);
                for (auto& c : poly) observe_ext(ch, c);
                a0 = challenger_sample_ext(ch);
                a1 = zero;
                alphas.push_back(a0);
                claim = poly_eval(poly, a0);
                PA = PA * (pt_a * a0 + (one - pt_a) * (one - a0));
            }
            if (pre) {                                       // the armed pass starts now
                uint32_t words[8];
                memcpy(words, a0.c, 16); memcpy(words + 4, a1.c, 16);
                gate.open(pre_ticket, words, 8);
            }
        }
        // the layer's last fold left one row per interaction: dense over 2^niv on the host
        std::vector<Ext> tn0(simd ? 1 : W, kb::ext_zero()), td0(simd ? 1 : W, one), tn1(simd ? 1 : W, kb::ext_zero()), td1(simd ? 1 : W, one);
        const size_t soa_stride = (((size_t)W + 15) / 16) * 16 + 16;
        std::vector<uint32_t> soa[2];
        {
            std::vector<size_t> so(K + 1, 0);
            for (uint32_t i = 0; i < K; i++) so[i + 1] = so[i] + (rows_at(info[int_chip[i]].rows, v + 1) ? 1 : 0);
            final_rows_total = so[K];
            const int cur = (int)((((v + 1) / 2)) & 1) ^ 1;  // folding passes of the layer: ceil(v / 2); they alternate scratch[0], [1], ...
            std::vector<Ext> host(std::max<size_t>(final_rows_total, 1) * 4);
            if (direct_final) { SP1HIP_TRY(mb.wait_next(host.data(), final_rows_total * 16, 4)); rsync.pending = false; }
            else SP1HIP_TRY(mb.fetch(scratch[cur].p, final_rows_total * 16, host.data()));
            if (simd) {                                      // coefficient planes: table w, plane k at (4 w + k) stride; all (0, 1) first
                soa[0].assign(16 * soa_stride, 0u);
                soa[1].assign(16 * soa_stride, 0u);
                for (int b = 0; b < 2; b++)
                    for (int w = 1; w < 4; w += 2) std::fill_n(soa[b].begin() + (size_t)(4 * w) * soa_stride, soa_stride, one.c[0]);
                for (uint32_t i = 0; i < K; i++) {
                    if (so[i + 1] == so[i]) continue;
                    for (int w = 0; w < 4; w++)
                        for (int k = 0; k < 4; k++) soa[0][(size_t)(4 * w + k) * soa_stride + i] = host[4 * so[i] + w].c[k];
                }
            } else
            for (uint32_t i = 0; i < K; i++) {
                if (so[i + 1] == so[i]) continue;            // chip without rows: stays (0, 1)
                const size_t base = 4 * so[i];
                tn0[i] = host[base]; td0[i] = host[base + 1]; tn1[i] = host[base + 2]; td1[i] = host[base + 3];
            }
        }
        if (gkr_debug) { const auto now = std::chrono::steady_clock::now(); dbg_rows += std::chrono::duration<double, std::milli>(now - dbg_t).count(); dbg_t = now; }
        // interaction-variable rounds on the host (InteractionLayer, logup_poly.rs:L240-L316); eq_adjustment = PA.
        // The eq table is never folded: after binding the last j variables it is eq_scale x the Lagrange table of the
        // first niv - j coordinates (an intermediate table of the construction above), and a Lagrange table sums to one,
        // which gives the padding entries' share of both sums without visiting them. The sums and the folds of a round
        // run on the helper threads (host_par.hpp); the tables ping-pong because a parallel fold cannot be in place.
        Ext eq_scale = one;
        Poly4 poly{};
        Ext alpha_r = zero;
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
            const int nparts_max = simd ? 1 : par.threads();
            for (int q = 0; q < nparts_max; q++) parts[q] = Part{kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
            if (simd) {
                uint32_t o6[6][4];
                gkr_host_round_sums(soa[side].data(), soa_stride, eq_soa[niv - j].data(), eq_plane_stride(niv - j), real_pairs, o6);
                Ext* dstp[6] = {&parts[0].x0, &parts[0].y0, &parts[0].xh, &parts[0].yh, &parts[0].e0, &parts[0].es};
                for (int q = 0; q < 6; q++) memcpy(dstp[q]->c, o6[q], 16);
            } else
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
            if (simd) {
                gkr_host_round_fold(soa[side].data(), soa[side ^ 1].data(), soa_stride, real_pairs, alpha_r.c);
                if (real_pairs < half)                       // the entry behind the last real one is the padding fraction (0, 1) again
                    for (int w = 0; w < 4; w++)
                        for (int k = 0; k < 4; k++) soa[side ^ 1][(size_t)(4 * w + k) * soa_stride + real_pairs] = (w & 1) && k == 0 ? one.c[0] : 0u;
            } else
            par.run(real_pairs, 48, [&](int, size_t kb0, size_t ke) {
                for (size_t k = kb0; k < ke; k++) {
                    o0[k] = n0[2 * k] + alpha_r * (n0[2 * k + 1] - n0[2 * k]); o1[k] = d0[2 * k] + alpha_r * (d0[2 * k + 1] - d0[2 * k]);
                    o2[k] = n1[2 * k] + alpha_r * (n1[2 * k + 1] - n1[2 * k]); o3[k] = d1[2 * k] + alpha_r * (d1[2 * k + 1] - d1[2 * k]);
                }
            });
            if (!simd && real_pairs < half) { o0[real_pairs] = kb::ext_zero(); o1[real_pairs] = one; o2[real_pairs] = kb::ext_zero(); o3[real_pairs] = one; }
            real = real_pairs;
            side ^= 1;
        }
        Ext fin_n0 = (*tab[side][0])[0], fin_d0 = (*tab[side][1])[0], fin_n1 = (*tab[side][2])[0], fin_d1 = (*tab[side][3])[0];
        if (simd) {
            Ext* f[4] = {&fin_n0, &fin_d0, &fin_n1, &fin_d1};
            for (int w = 0; w < 4; w++)
                for (int k = 0; k < 4; k++) f[w]->c[k] = soa[side][(size_t)(4 * w + k) * soa_stride];
        }
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
        par.park();
    }

    if (gkr_debug) fprintf(stderr, "[sp1hip gkr]   of which row-variable rounds %.3f ms, interaction-variable rounds (host) %.3f ms\n", dbg_rows, dbg_int);
    if (gkr_debug) fprintf(stderr, "[sp1hip gkr]   %d hand-overs inside the layers: waiting for the sums %.3f ms, closed forms + transcript %.3f ms, launch calls %.3f ms\n",
                           dbg_passes, dbg_wait, dbg_host, dbg_launch);
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
        hipLaunchKernelGGL(open_sum_kernel, dim3(((uint32_t)total_cols * 4 + 63) / 64), dim3(256), 0, s, d_part.u32(), chunks,
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
