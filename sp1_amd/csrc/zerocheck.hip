// sp1_amd/csrc/zerocheck.hip — zerocheck sumcheck over AIR constraints on gfx950 (SURVEY §8 a9–a12).
//
//   zc_sum_kernel            `ZerocheckCpuProver::sum_as_poly_in_last_variable` + `increment_y_values`
//                            /root/reference/crates/hypercube/src/prover/zerocheck/sum_as_poly.rs:L53-L181,L355-L440
//                            with `ConstraintSumcheckFolder::assert_zero` (/root/reference/crates/hypercube/src/folder.rs:L276-L323)
//   zc_fix_kernel            `zerocheck_fix_last_variable` -> `mle_fix_last_variable`
//                            /root/reference/crates/hypercube/src/prover/zerocheck/fix_last_variable.rs:L8-L62,
//                            /root/reference/slop/crates/multilinear/src/restrict.rs:L11-L58
//   host driver              `ShardProver::zerocheck` (/root/reference/crates/hypercube/src/prover/shard.rs:L474-L646),
//                            `reduce_sumcheck_to_evaluation` (/root/reference/slop/crates/sumcheck/src/prover.rs:L13-L96),
//                            univariate assembly sum_as_poly.rs:L187-L287, `VirtualGeq`
//                            (/root/reference/slop/crates/multilinear/src/virtual_geq.rs:L12-L99)
//
// Constraints are data: an SSA program per chip (include/sp1hip.h, sp1_amd/air.py). The host
// linear-scan allocates registers; the kernel is a register machine per row pair — one lane = one
// pair of adjacent rows, evaluated at the interpolation nodes t = 0, 2, 4 (leaf = row0 + t (row1 -
// row0)), `acc += alpha_pow[k] * reg` per assert, plus the GKR-opening batching term, times eq(zeta',
// pair), block-reduced to three extension sums. Traces are column-major, so every leaf load of a
// wave is one coalesced 256 B run; the program, alpha/gkr powers and publics are wave-uniform
// scalar loads. Round 0 works on base-field words, later rounds on extension words (4 sub-columns
// per column). The register file lives in per-lane scratch (runtime-indexed); an ahead-of-time
// specialised kernel per chip (registers in VGPRs, no decode) is the planned replacement.
#include <algorithm>
#include <array>
#include <cstring>
#include <memory>
#include <vector>

#include "device_ctx.hpp"
#include "round_sync.hpp"

namespace sp1hip {

enum ZcOp : uint32_t { ZC_LOAD_MAIN = 0, ZC_LOAD_PREP = 1, ZC_CONST = 2, ZC_PUBLIC = 3, ZC_ADD = 4, ZC_SUB = 5, ZC_MUL = 6,
                       ZC_NEG = 7, ZC_ASSERT_ZERO = 8 };

// One chip of the current round (device array; every field is wave-uniform in the kernels).
struct ZcDesc {
    const uint32_t* prog;        // [n_instr][4]: op | flags, dst, a, b (register-allocated)
    const uint32_t* main;        // column-major; round 0: [rows x main_w] base words, later [rows x 4 main_w]
    const uint32_t* prep;
    const uint32_t* alpha_pows;  // [num_constraints][4]
    const uint32_t* gkr_pows;    // [main_w + prep_w][4]
    uint32_t n_instr, main_w, prep_w, rows;
    uint32_t block_start, n_blocks;
    uint32_t alpha_off;          // index of this chunk's first constraint in alpha_pows
    uint32_t flags;              // bit 0: first chunk of its chip (owns the round-0 GKR-only pass)
};

// Blocks of one chip (all its chunks are contiguous) for the reduction, plus the eq entry it needs.
struct ZcChipRange {
    uint32_t block_start, n_blocks, th, pad;
};

struct ZcFixDesc {
    const uint32_t* in;
    uint32_t* out;
    uint32_t rows, width, block_start, n_blocks;
};

// ---- K = base word (round 0) or extension element (later rounds)
template <bool FIRST> struct KT;
template <> struct KT<true> {
    using T = uint32_t;
    static __device__ __forceinline__ T zero() { return 0u; }
    static __device__ __forceinline__ T from_f(uint32_t x) { return x; }
    static __device__ __forceinline__ T add(T a, T b) { return kb::add(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return kb::sub(a, b); }
    static __device__ __forceinline__ T mul(T a, T b) { return kb::mul(a, b); }
    static __device__ __forceinline__ kb::Ext scale(const kb::Ext& e, T k) { return kb::ext_mul_base(e, k); }
    static __device__ __forceinline__ kb::Ext to_ext(T k) { return kb::ext_from_base(k); }
    static __device__ __forceinline__ T load(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t r) {
        return tbl[(size_t)col * rows + r];
    }
};
template <> struct KT<false> {
    using T = kb::Ext;
    static __device__ __forceinline__ T zero() { return kb::ext_zero(); }
    static __device__ __forceinline__ T from_f(uint32_t x) { return kb::ext_from_base(x); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return kb::ext_add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return kb::ext_sub(a, b); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return kb::ext_mul(a, b); }
    static __device__ __forceinline__ kb::Ext scale(const kb::Ext& e, const T& k) { return kb::ext_mul(k, e); }   // e is wave-uniform
    static __device__ __forceinline__ kb::Ext to_ext(const T& k) { return k; }
    static __device__ __forceinline__ T load(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t r) {
        T v;
#pragma unroll
        for (int k = 0; k < 4; k++) v.c[k] = tbl[((size_t)col * 4 + k) * rows + r];
        return v;
    }
};

__device__ __forceinline__ kb::Ext load_ext_aos(const uint32_t* p, uint32_t i) {
    return kb::Ext{{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}};
}

// value of column `col` at node t in {0, 2, 4} for row pair i
template <bool FIRST>
__device__ __forceinline__ typename KT<FIRST>::T leaf(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t i, int t) {
    using K = KT<FIRST>;
    typename K::T r0 = K::load(tbl, col, rows, 2 * i);
    if (t == 0) return r0;
    typename K::T r1 = (2 * i + 1 < rows) ? K::load(tbl, col, rows, 2 * i + 1) : K::zero();
    typename K::T slope = K::sub(r1, r0);
    typename K::T s2 = K::add(slope, slope);
    if (t == 2) return K::add(s2, r0);
    return K::add(K::add(s2, s2), r0);
}

// ---- register files ------------------------------------------------------------------------------
// The program is wave-uniform, so register numbers are SGPR values. For up to 32 registers the file is
// kept in VGPRs as 16/32-wide vectors indexed with a uniform index (GPR-index mode, s_set_gpr_idx_on):
// no memory traffic per interpreted instruction. Larger programs fall back to per-lane scratch.
typedef uint32_t v32u __attribute__((ext_vector_type(32)));
typedef uint32_t v16u __attribute__((ext_vector_type(16)));

template <bool FIRST, int MAXR> struct RegFile {
    typename KT<FIRST>::T r[MAXR];
    __device__ __forceinline__ typename KT<FIRST>::T get(uint32_t i) const { return r[i]; }
    __device__ __forceinline__ void set(uint32_t i, const typename KT<FIRST>::T& v) { r[i] = v; }
};
#define SP1HIP_VREGFILE(N, V)                                                                                       \
    template <> struct RegFile<true, N> {                                                                            \
        V r;                                                                                                         \
        __device__ __forceinline__ uint32_t get(uint32_t i) const { return r[i]; }                                  \
        __device__ __forceinline__ void set(uint32_t i, uint32_t v) { r[i] = v; }                                   \
    };                                                                                                               \
    template <> struct RegFile<false, N> {                                                                           \
        V c0, c1, c2, c3;                                                                                            \
        __device__ __forceinline__ kb::Ext get(uint32_t i) const { return kb::Ext{{c0[i], c1[i], c2[i], c3[i]}}; }   \
        __device__ __forceinline__ void set(uint32_t i, const kb::Ext& v) {                                          \
            c0[i] = v.c[0]; c1[i] = v.c[1]; c2[i] = v.c[2]; c3[i] = v.c[3];                                          \
        }                                                                                                            \
    };
SP1HIP_VREGFILE(16, v16u)
SP1HIP_VREGFILE(32, v32u)

// MAXR == 0: the file lives in LDS, register i of a lane at slot i * 256 + lane (16 B slots for extension values: one
// ds_read_b128 / ds_write_b128 per access, conflict-free). Indexing a VGPR vector with a wave-uniform index costs an
// s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off triple per word — ~36 instructions of pure register traffic around a
// 12-instruction extension add; the LDS file makes an interpreted op cost its arithmetic plus three LDS accesses,
// and leaves the VGPRs to the arithmetic (measured per-op cost: add 75 -> ~20 instructions, multiply 147 -> ~100).
template <> struct RegFile<true, 0> {
    uint32_t* base;                // this lane's slot of register 0
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return base[i * 256u]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) { base[i * 256u] = v; }
};
template <> struct RegFile<false, 0> {
    uint4* base;
    __device__ __forceinline__ kb::Ext get(uint32_t i) const { const uint4 v = base[i * 256u]; return kb::Ext{{v.x, v.y, v.z, v.w}}; }
    __device__ __forceinline__ void set(uint32_t i, const kb::Ext& v) { base[i * 256u] = make_uint4(v.c[0], v.c[1], v.c[2], v.c[3]); }
};

constexpr uint32_t ZC_GKR_FLAG = 0x100u;   // set by the host on the first load of each column
constexpr uint32_t ZC_TOUCH = 9;           // pseudo-op: column never loaded by the constraints (GKR term only)
constexpr uint32_t ZC_CHUNK_LIMIT = 96;    // target instructions per chunk (host-side program splitting)
constexpr uint32_t ZC_LDS_PROG_MAX = 3072; // instructions staged in LDS (48 KiB); longer programs read global memory

// One pass of the program at node t. With `gkr`, the first load of every column also accumulates
// gkr_pow[column] * value into *g (main columns first, then preprocessed): the batching term costs no
// extra loads. `prog` points to LDS (or global memory for very long programs).
template <bool FIRST, int MAXR>
__device__ __forceinline__ kb::Ext run_program(RegFile<FIRST, MAXR>& reg, const uint4* prog, const ZcDesc& d,
                                               const uint32_t* __restrict__ publics, uint32_t i, int t, const bool gkr, kb::Ext* g) {
    using K = KT<FIRST>;
    kb::Ext acc = kb::ext_zero();
    uint32_t ci = 0;
    for (uint32_t k = 0; k < d.n_instr; k++) {
        const uint4 w = prog[k];     // wave-uniform: decode once, keep the fields in SGPRs
        const uint32_t opw = __builtin_amdgcn_readfirstlane(w.x), dst = __builtin_amdgcn_readfirstlane(w.y);
        const uint32_t x = __builtin_amdgcn_readfirstlane(w.z), y = __builtin_amdgcn_readfirstlane(w.w);
        switch (opw & 0xffu) {
            case ZC_LOAD_MAIN: {
                typename K::T v = leaf<FIRST>(d.main, x, d.rows, i, t);
                if (gkr && (opw & ZC_GKR_FLAG)) *g = kb::ext_add(*g, K::scale(load_ext_aos(d.gkr_pows, x), v));
                reg.set(dst, v);
                break;
            }
            case ZC_LOAD_PREP: {
                typename K::T v = leaf<FIRST>(d.prep, x, d.rows, i, t);
                if (gkr && (opw & ZC_GKR_FLAG)) *g = kb::ext_add(*g, K::scale(load_ext_aos(d.gkr_pows, d.main_w + x), v));
                reg.set(dst, v);
                break;
            }
            case ZC_TOUCH:
                if (gkr) {
                    typename K::T v = leaf<FIRST>(y ? d.prep : d.main, x, d.rows, i, t);
                    *g = kb::ext_add(*g, K::scale(load_ext_aos(d.gkr_pows, (y ? d.main_w : 0u) + x), v));
                }
                break;
            case ZC_CONST: reg.set(dst, K::from_f(x)); break;               // host pre-converts to Montgomery
            case ZC_PUBLIC: reg.set(dst, K::from_f(publics[x])); break;
            case ZC_ADD: reg.set(dst, K::add(reg.get(x), reg.get(y))); break;
            case ZC_SUB: reg.set(dst, K::sub(reg.get(x), reg.get(y))); break;
            case ZC_MUL: reg.set(dst, K::mul(reg.get(x), reg.get(y))); break;
            case ZC_NEG: reg.set(dst, K::sub(K::zero(), reg.get(x))); break;
            default: acc = kb::ext_add(acc, K::scale(load_ext_aos(d.alpha_pows, d.alpha_off + ci++), reg.get(x))); break;  // ASSERT_ZERO
        }
    }
    return acc;
}

__device__ __forceinline__ uint32_t zc_wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, __shfl_xor(v, off));
    return v;
}

// last descriptor whose block_start <= bid (binary search; everything stays wave-uniform)
__device__ __forceinline__ ZcDesc zc_find_desc(const ZcDesc* __restrict__ descs, int n, uint32_t bid) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (__builtin_amdgcn_readfirstlane(descs[mid].block_start) <= bid) lo = mid; else hi = mid - 1;
    }
    return descs[lo];
}

// One launch per sumcheck round covers EVERY chip and the three interpolation nodes:
//   blockIdx.x = 3 b + p -> (chip, block b of 256 row pairs), pass p (node t = 2p).
// A pass-p workgroup writes two extension partial sums [A | B] (8 words):
//   round 0 :  p=0: A = sum eq g(0), B = sum eq g(2)   (GKR batching term only; constraints vanish at 0)
//              p=1: A = sum eq C(2)                     p=2: A = sum eq C(4)
//   later   :  p=0: A = sum eq C(0), B = sum eq g(0)    p=1: A = sum eq C(2), B = sum eq g(2)    p=2: A = sum eq C(4)
// g(4) = 2 g(2) - g(0) is linear, so the three nodes can run in different workgroups and the late, tiny
// rounds (latency-bound: one wave interprets the whole program serially) run all chips and nodes at once.
template <bool FIRST, int MAXR>
__global__ __launch_bounds__(256) void zc_round_kernel(const ZcDesc* __restrict__ descs, int n_descs,
                                                       const uint32_t* __restrict__ eq, uint32_t eq_len,
                                                       const uint32_t* __restrict__ publics, uint32_t* __restrict__ partial,
                                                       uint32_t rf_off) {
    using K = KT<FIRST>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* red = lds;                                   // [4][8] reduction scratch
    uint4* lprog = reinterpret_cast<uint4*>(lds + 32);
    RegFile<FIRST, MAXR> reg;
    if constexpr (MAXR == 0) reg.base = reinterpret_cast<decltype(reg.base)>(lds + rf_off) + threadIdx.x;   // LDS file behind the program
    // the three nodes of one block of row pairs are CONSECUTIVE workgroups: they are dispatched together (to
    // different XCDs), so the second and third read of the same table rows hit the memory-side cache instead of HBM
    const uint32_t bid = blockIdx.x / 3u;
    const int pass = (int)(blockIdx.x - 3u * bid);
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const bool in_lds = d.n_instr <= ZC_LDS_PROG_MAX;
    if (in_lds) {
        const uint4* src = reinterpret_cast<const uint4*>(d.prog);
        for (uint32_t k = threadIdx.x; k < d.n_instr; k += 256) lprog[k] = src[k];
        __syncthreads();
    }
    const uint4* prog = in_lds ? lprog : reinterpret_cast<const uint4*>(d.prog);
    const uint32_t terms = (d.rows + 1) / 2;
    kb::Ext sa = kb::ext_zero(), sb = kb::ext_zero();
    for (uint32_t i = (bid - d.block_start) * 256u + threadIdx.x; i < terms; i += d.n_blocks * 256u) {
        kb::Ext va = kb::ext_zero(), vb = kb::ext_zero();
        if (FIRST && pass == 0) {
            if (d.flags & 1u)
            for (uint32_t c = 0; c < d.main_w; c++) {
                const kb::Ext pw = load_ext_aos(d.gkr_pows, c);
                va = kb::ext_add(va, K::scale(pw, leaf<FIRST>(d.main, c, d.rows, i, 0)));
                vb = kb::ext_add(vb, K::scale(pw, leaf<FIRST>(d.main, c, d.rows, i, 2)));
            }
            if (d.flags & 1u)
            for (uint32_t c = 0; c < d.prep_w; c++) {
                const kb::Ext pw = load_ext_aos(d.gkr_pows, d.main_w + c);
                va = kb::ext_add(va, K::scale(pw, leaf<FIRST>(d.prep, c, d.rows, i, 0)));
                vb = kb::ext_add(vb, K::scale(pw, leaf<FIRST>(d.prep, c, d.rows, i, 2)));
            }
        } else {
            va = run_program<FIRST, MAXR>(reg, prog, d, publics, i, 2 * pass, !FIRST && pass < 2, &vb);
        }
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        sa = kb::ext_add(sa, kb::ext_mul(va, e));
        sb = kb::ext_add(sb, kb::ext_mul(vb, e));
    }
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = sa.c[k]; v[4 + k] = sb.c[k]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = zc_wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) red[wave * 8 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const uint32_t k = threadIdx.x;
        partial[((size_t)bid * 3 + pass) * 8 + k] = kb::add(kb::add(red[k], red[8 + k]), kb::add(red[16 + k], red[24 + k]));
    }
}

// One workgroup per chip: sums its workgroups' partials and forms (y0, y2, y4, eq[th]) -> out[chip][16].
template <bool FIRST>
__global__ __launch_bounds__(256) void zc_reduce_kernel(const ZcChipRange* __restrict__ ranges, const uint32_t* __restrict__ partial,
                                                        const uint32_t* __restrict__ eq, uint32_t eq_len,
                                                        uint32_t* __restrict__ out) {
    __shared__ uint32_t acc[10][24];
    const ZcChipRange d = ranges[blockIdx.x];
    const uint32_t word = threadIdx.x % 24, grp = threadIdx.x / 24;   // 10 groups of 24 words (3 passes x 8)
    if (grp < 10) {
        uint32_t a = 0;
        for (uint32_t b = grp; b < d.n_blocks; b += 10) a = kb::add(a, partial[((size_t)(d.block_start + b)) * 24 + word]);
        acc[grp][word] = a;
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        uint32_t a = 0;
        for (int g = 0; g < 10; g++) a = kb::add(a, acc[g][threadIdx.x]);
        acc[0][threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const uint32_t k = threadIdx.x;
        // S[p][0..4) = A of pass p, S[p][4..8) = B of pass p
        const uint32_t A0 = acc[0][k], B0 = acc[0][4 + k], A1 = acc[0][8 + k], B1 = acc[0][12 + k], A2 = acc[0][16 + k];
        uint32_t y0, y2, y4;
        if (FIRST) {       // g0 = A0, g2 = B0, C(2) = A1, C(4) = A2
            y0 = A0;
            y2 = kb::add(A1, B0);
            y4 = kb::add(A2, kb::sub(kb::add(B0, B0), A0));
        } else {           // C(0) = A0, g0 = B0, C(2) = A1, g2 = B1, C(4) = A2
            y0 = kb::add(A0, B0);
            y2 = kb::add(A1, B1);
            y4 = kb::add(A2, kb::sub(kb::add(B1, B1), B0));
        }
        uint32_t* o = out + (size_t)blockIdx.x * 16;
        o[k] = y0; o[4 + k] = y2; o[8 + k] = y4;
        o[12 + k] = d.th < eq_len ? eq[(size_t)k * eq_len + d.th] : 0u;
    }
}

// out[i][c] = x + alpha (y - x), x = row 2i, y = row 2i + 1 (zero beyond the real rows); out is an ext table.
// One launch per round for every table of every chip.
template <bool FIRST>
__global__ __launch_bounds__(256) void zc_fix_kernel(const ZcFixDesc* __restrict__ descs, int n_descs, kb::Ext alpha) {
    using K = KT<FIRST>;
    int k = 0;
    for (int i = 1; i < n_descs; i++)
        if (__builtin_amdgcn_readfirstlane(descs[i].block_start) <= blockIdx.x) k = i;
    const ZcFixDesc d = descs[k];
    const uint32_t out_rows = (d.rows + 1) / 2;
    const size_t t = (size_t)(blockIdx.x - d.block_start) * 256u + threadIdx.x;
    if (t >= (size_t)out_rows * d.width) return;
    const uint32_t c = (uint32_t)(t / out_rows), i = (uint32_t)(t % out_rows);
    typename K::T x = K::load(d.in, c, d.rows, 2 * i);
    typename K::T y = (2 * i + 1 < d.rows) ? K::load(d.in, c, d.rows, 2 * i + 1) : K::zero();
    const kb::Ext r = kb::ext_add(K::scale(alpha, K::sub(y, x)), K::to_ext(x));
#pragma unroll
    for (int q = 0; q < 4; q++) d.out[((size_t)c * 4 + q) * out_rows + i] = r.c[q];
}

struct ZcGatherDesc { const uint32_t* src; uint32_t n_words, dst_off; };
__global__ __launch_bounds__(256) void zc_gather_kernel(const ZcGatherDesc* __restrict__ descs, uint32_t* __restrict__ out) {
    const ZcGatherDesc d = descs[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < d.n_words; i += 256) out[d.dst_off + i] = d.src[i];
}

// ------------------------------------------------------------------------------------------ host side
using Ext = kb::Ext;
static Ext operator+(const Ext& a, const Ext& b) { return kb::ext_add(a, b); }
static Ext operator-(const Ext& a, const Ext& b) { return kb::ext_sub(a, b); }
static Ext operator*(const Ext& a, const Ext& b) { return kb::ext_mul(a, b); }
static Ext ext_c(uint32_t canonical) { return kb::ext_from_base(kb::to_monty(canonical)); }

using UniPoly = std::vector<Ext>;
static Ext uni_eval(const UniPoly& p, const Ext& x) {
    Ext acc = kb::ext_zero();
    for (size_t i = p.size(); i-- > 0;) acc = acc * x + p[i];
    return acc;
}
static UniPoly uni_add(const UniPoly& a, const UniPoly& b) {
    UniPoly r(std::max(a.size(), b.size()), kb::ext_zero());
    for (size_t i = 0; i < r.size(); i++) r[i] = (i < a.size() ? a[i] : kb::ext_zero()) + (i < b.size() ? b[i] : kb::ext_zero());
    return r;
}
static UniPoly uni_scale(UniPoly a, const Ext& k) { for (auto& c : a) c = c * k; return a; }
// Lagrange interpolation, same operation order as slop_algebra::interpolate_univariate_polynomial
static UniPoly interpolate(const std::vector<Ext>& xs, const std::vector<Ext>& ys) {
    UniPoly result{kb::ext_zero()};
    for (size_t i = 0; i < xs.size(); i++) {
        Ext den = kb::ext_one();
        UniPoly num{ys[i]};
        for (size_t j = 0; j < xs.size(); j++) {
            if (j == i) continue;
            den = den * (xs[i] - xs[j]);
            UniPoly shifted{kb::ext_zero()};
            shifted.insert(shifted.end(), num.begin(), num.end());
            num = uni_add(shifted, uni_scale(num, kb::ext_zero() - xs[j]));
        }
        result = uni_add(result, uni_scale(num, kb::ext_inv(den)));
    }
    return result;
}

struct VGeq {
    uint32_t threshold;
    Ext geq_c, eq_c;
    VGeq fix(const Ext& alpha) const {
        VGeq r;
        r.threshold = threshold >> 1;
        r.geq_c = geq_c;
        r.eq_c = (threshold & 1) == 0 ? (kb::ext_one() - alpha) * eq_c : alpha * (eq_c + geq_c) - geq_c;
        return r;
    }
    Ext at(size_t idx) const {
        if (idx < threshold) return kb::ext_zero();
        if (idx == threshold) return eq_c + geq_c;
        return geq_c;
    }
};

struct DevBuf {
    void* p = nullptr;
    hipStream_t s = nullptr;
    size_t n = 0;
    int alloc(size_t bytes, hipStream_t stream) {
        s = stream;
        n = bytes;
        return arena_alloc(&p, bytes, stream);
    }
    void release() { arena_free(p, n, s); p = nullptr; }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    uint32_t* u32() const { return (uint32_t*)p; }
};

struct Chunk {
    std::vector<uint32_t> prog;   // allocated [n][4]
    uint32_t n_regs = 1, alpha_off = 0;
};

struct ChipState {
    const sp1hip_zc_chip_t* in;
    std::vector<uint32_t> prog;     // allocated [n][4]
    uint32_t n_regs = 1;
    std::vector<Ext> alpha_pows, gkr_pows;
    std::vector<Chunk> chunks;
    std::vector<uint32_t> chunk_off;   // offset (in instructions) of each chunk inside d_prog
    size_t off_prog = 0, off_alpha = 0, off_gkr = 0;     // word offsets into the call's single constant blob
    const uint32_t* p_prog = nullptr;
    const uint32_t* p_alpha = nullptr;
    const uint32_t* p_gkr = nullptr;
    std::unique_ptr<DevBuf> main_buf, prep_buf;   // ext tables of later rounds
    const uint32_t* d_main = nullptr;
    const uint32_t* d_prep = nullptr;
    uint64_t rows = 0;
    uint32_t num_vars = 0;
    Ext eq_adj, pad_adj;
    VGeq vgeq;
    UniPoly uni;
};

// linear-scan register allocation of the SSA program (host)
static int allocate_registers(const uint32_t* ssa, uint32_t n, std::vector<uint32_t>* out, uint32_t* n_regs) {
    std::vector<int> last_use(n, -1);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        SP1HIP_REQUIRE(op <= ZC_ASSERT_ZERO, "bad opcode in constraint program");
        if (op == ZC_ADD || op == ZC_SUB || op == ZC_MUL) {
            SP1HIP_REQUIRE(a < k && b < k, "constraint program is not in SSA order");
            last_use[a] = (int)k;
            last_use[b] = (int)k;
        } else if (op == ZC_NEG || op == ZC_ASSERT_ZERO) {
            SP1HIP_REQUIRE(a < k, "constraint program is not in SSA order");
            last_use[a] = (int)k;
        }
    }
    std::vector<uint32_t> free_regs, reg_of(n, 0xffffffffu);
    uint32_t regs = 0;
    out->assign((size_t)n * 4, 0);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        const bool bin = op == ZC_ADD || op == ZC_SUB || op == ZC_MUL, un = op == ZC_NEG || op == ZC_ASSERT_ZERO;
        uint32_t ra = a, rb = b;
        if (bin || un) ra = reg_of[a];
        if (bin) rb = reg_of[b];
        if (bin || un) {
            if (last_use[a] == (int)k && reg_of[a] != 0xffffffffu) { free_regs.push_back(reg_of[a]); reg_of[a] = 0xffffffffu; }
            if (bin && b != a && last_use[b] == (int)k && reg_of[b] != 0xffffffffu) { free_regs.push_back(reg_of[b]); reg_of[b] = 0xffffffffu; }
        }
        uint32_t dst = 0;
        if (op != ZC_ASSERT_ZERO) {
            if (!free_regs.empty()) { dst = free_regs.back(); free_regs.pop_back(); }
            else dst = regs++;
            if (last_use[k] >= 0) reg_of[k] = dst;
            else free_regs.push_back(dst);      // dead value
        }
        uint32_t* o = out->data() + (size_t)k * 4;
        o[0] = op; o[1] = dst; o[2] = ra; o[3] = rb;
        if (op == ZC_CONST) o[2] = kb::to_monty(a % kb::P);
    }
    *n_regs = regs ? regs : 1;
    return SP1HIP_SUCCESS;
}

// Splits the SSA program into self-contained chunks at assert boundaries (each chunk re-emits the
// dependency cone of its asserts, at most ~`limit` instructions unless a single cone is larger). Chunks
// are independent workgroups on the GPU: wide chips get parallelism across constraints, which is what
// keeps the late, tiny sumcheck rounds from being one wave interpreting thousands of instructions
// serially (cf. the reference's chunked bytecode, /root/reference/sp1-gpu/crates/air/src/ir/bytecode.rs:L27-L110).
static int build_chunks(const uint32_t* ssa, uint32_t n, uint32_t main_w, uint32_t prep_w, uint32_t limit,
                        std::vector<Chunk>* out) {
    std::vector<uint32_t> stamp(n, 0xffffffffu);
    std::vector<uint32_t> members, asserts, stack;
    uint32_t chunk_id = 0, assert_index = 0, first_assert = 0;
    auto flush = [&]() -> int {
        if (asserts.empty()) return SP1HIP_SUCCESS;
        std::sort(members.begin(), members.end());
        std::vector<uint32_t> renum(n, 0), sub;
        // interleave: every member instruction in original order, asserts after their operand exists
        std::vector<std::pair<uint32_t, bool>> order;   // (ssa index, is_assert)
        for (uint32_t m : members) order.push_back({m, false});
        for (uint32_t a : asserts) order.push_back({a, true});
        std::sort(order.begin(), order.end());
        uint32_t next = 0;
        for (auto& o : order) {
            const uint32_t k = o.first, op = ssa[3 * k];
            uint32_t a = ssa[3 * k + 1], b = ssa[3 * k + 2];
            if (op == ZC_ADD || op == ZC_SUB || op == ZC_MUL) { a = renum[a]; b = renum[b]; }
            else if (op == ZC_NEG || op == ZC_ASSERT_ZERO) a = renum[a];
            renum[k] = next++;
            sub.insert(sub.end(), {op, a, b});
        }
        Chunk c;
        c.alpha_off = first_assert;
        SP1HIP_TRY(allocate_registers(sub.data(), (uint32_t)(sub.size() / 3), &c.prog, &c.n_regs));
        out->push_back(std::move(c));
        members.clear();
        asserts.clear();
        chunk_id++;
        return SP1HIP_SUCCESS;
    };
    for (uint32_t k = 0; k < n; k++) {
        if (ssa[3 * k] != ZC_ASSERT_ZERO) continue;
        // new nodes this assert would add to the current chunk
        std::vector<uint32_t> fresh;
        stack.assign(1, ssa[3 * k + 1]);
        while (!stack.empty()) {
            const uint32_t v = stack.back();
            stack.pop_back();
            if (stamp[v] == chunk_id) continue;
            stamp[v] = chunk_id;
            fresh.push_back(v);
            const uint32_t op = ssa[3 * v];
            if (op == ZC_ADD || op == ZC_SUB || op == ZC_MUL) { stack.push_back(ssa[3 * v + 1]); stack.push_back(ssa[3 * v + 2]); }
            else if (op == ZC_NEG) stack.push_back(ssa[3 * v + 1]);
        }
        if (!asserts.empty() && members.size() + fresh.size() + asserts.size() + 1 > limit) {
            for (uint32_t v : fresh) stamp[v] = 0xffffffffu;     // undo, close the chunk, retry in a new one
            SP1HIP_TRY(flush());
            k--;
            continue;
        }
        if (asserts.empty()) first_assert = assert_index;
        members.insert(members.end(), fresh.begin(), fresh.end());
        asserts.push_back(k);
        assert_index++;
    }
    SP1HIP_TRY(flush());
    // GKR visits: the first load of each column, in chunk order, carries the flag; columns no constraint
    // reads get TOUCH pseudo-instructions in extra chunks
    std::vector<bool> seen_m(main_w, false), seen_p(prep_w, false);
    for (auto& c : *out)
        for (size_t k = 0; k < c.prog.size() / 4; k++) {
            uint32_t* o = c.prog.data() + 4 * k;
            if (o[0] == ZC_LOAD_MAIN && !seen_m[o[2]]) { seen_m[o[2]] = true; o[0] |= ZC_GKR_FLAG; }
            if (o[0] == ZC_LOAD_PREP && !seen_p[o[2]]) { seen_p[o[2]] = true; o[0] |= ZC_GKR_FLAG; }
        }
    Chunk touch;
    auto push_touch = [&](uint32_t col, uint32_t is_prep) {
        touch.prog.insert(touch.prog.end(), {ZC_TOUCH, 0u, col, is_prep});
        if (touch.prog.size() / 4 >= limit) { out->push_back(touch); touch.prog.clear(); }
    };
    for (uint32_t c = 0; c < main_w; c++) if (!seen_m[c]) push_touch(c, 0);
    for (uint32_t c = 0; c < prep_w; c++) if (!seen_p[c]) push_touch(c, 1);
    if (!touch.prog.empty()) out->push_back(touch);
    if (out->empty()) { Chunk e; e.prog = {ZC_TOUCH, 0u, 0u, 2u}; out->push_back(e); }   // no constraints, no columns
    return SP1HIP_SUCCESS;
}

// host evaluation of the program on an all-zero row (padded_row_adjustment, shard.rs:L524-L536)
static Ext eval_zero_row(const ChipState& c, const uint32_t* publics) {
    const uint32_t n = (uint32_t)(c.prog.size() / 4);
    std::vector<uint32_t> reg(c.n_regs, 0);
    Ext acc = kb::ext_zero();
    uint32_t ci = 0;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = c.prog[4 * k] & 0xffu, dst = c.prog[4 * k + 1], x = c.prog[4 * k + 2], y = c.prog[4 * k + 3];
        switch (op) {
            case ZC_LOAD_MAIN: case ZC_LOAD_PREP: reg[dst] = 0; break;
            case ZC_TOUCH: break;
            case ZC_CONST: reg[dst] = x; break;
            case ZC_PUBLIC: reg[dst] = publics[x]; break;
            case ZC_ADD: reg[dst] = kb::add(reg[x], reg[y]); break;
            case ZC_SUB: reg[dst] = kb::sub(reg[x], reg[y]); break;
            case ZC_MUL: reg[dst] = kb::mul(reg[x], reg[y]); break;
            case ZC_NEG: reg[dst] = kb::neg(reg[x]); break;
            default: acc = acc + kb::ext_mul_base(c.alpha_pows[ci++], reg[x]); break;
        }
    }
    return acc;
}

template <bool FIRST>
static int launch_round(uint32_t max_regs, const ZcDesc* d_descs, int n_descs, uint32_t total_blocks, uint32_t max_instr,
                        const uint32_t* eq, uint32_t eq_len, const uint32_t* publics, uint32_t* partial, hipStream_t s) {
    const uint32_t staged = max_instr <= ZC_LDS_PROG_MAX ? max_instr : 0;
    const size_t lds = 32 * 4 + (size_t)staged * 16;
    dim3 grid(total_blocks * 3);          // workgroup 3 b + p = node p of block b
    // register file in LDS when it fits 64 KiB together with the staged program (two workgroups per CU at worst)
    const size_t rf_bytes = (size_t)max_regs * 256 * (FIRST ? 4 : 16);
    static const bool force_vgpr = [] { const char* e = getenv("SP1HIP_ZC_REGFILE"); return e && e[0] == 'v'; }();
    if (!force_vgpr && lds + rf_bytes <= 64 * 1024) {
        auto kern = zc_round_kernel<FIRST, 0>;
        if (lds + rf_bytes > 48 * 1024)
            SP1HIP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + rf_bytes)));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds + rf_bytes, s, d_descs, n_descs, eq, eq_len, publics, partial, (uint32_t)(lds / 4));
        SP1HIP_LAUNCH_CHECK();
        return SP1HIP_SUCCESS;
    }
    if (max_regs <= 16) hipLaunchKernelGGL((zc_round_kernel<FIRST, 16>), grid, dim3(256), lds, s, d_descs, n_descs, eq, eq_len, publics, partial, 0u);
    else if (max_regs <= 32) hipLaunchKernelGGL((zc_round_kernel<FIRST, 32>), grid, dim3(256), lds, s, d_descs, n_descs, eq, eq_len, publics, partial, 0u);
    else if (max_regs <= 256) hipLaunchKernelGGL((zc_round_kernel<FIRST, 256>), grid, dim3(256), lds, s, d_descs, n_descs, eq, eq_len, publics, partial, 0u);
    else if (max_regs <= 1024) hipLaunchKernelGGL((zc_round_kernel<FIRST, 1024>), grid, dim3(256), lds, s, d_descs, n_descs, eq, eq_len, publics, partial, 0u);
    else { set_error("constraint program needs %u live registers (max 1024)", max_regs); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

struct ByteOut {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void ext(const Ext& e) {
        for (int k = 0; k < 4; k++) { uint32_t c = kb::from_monty(e.c[k]); for (int i = 0; i < 4; i++) b.push_back((uint8_t)(c >> (8 * i))); }
    }
};

}  // namespace sp1hip

using namespace sp1hip;

struct sp1hip_challenger_s;
namespace sp1hip {
// transcript hooks implemented in prover.hip
void challenger_observe(sp1hip_challenger_t* ch, uint32_t x);
kb::Ext challenger_sample_ext(sp1hip_challenger_t* ch);
void challenger_restore(sp1hip_challenger_t* dst, const sp1hip_challenger_t* src);
}

static int zerocheck_prove_impl(const sp1hip_zc_chip_t* chips, int n_chips, int max_log_row_count,
                               const sp1hip_ext_t* h_zeta, const sp1hip_ext_t* h_openings, sp1hip_ext_t alpha_c,
                               sp1hip_ext_t gkr_c, const uint32_t* h_publics, int n_publics,
                               sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                               sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(chips && n_chips > 0 && h_zeta && h_openings && challenger && proof_len, "null argument");
    SP1HIP_REQUIRE(max_log_row_count >= 1 && max_log_row_count <= 30, "max_log_row_count out of range");
    SP1HIP_REQUIRE(h_publics || n_publics == 0, "null publics");
    const int L = max_log_row_count;
    size_t total_w = 0;
    for (int i = 0; i < n_chips; i++) {
        // a chip may have no constraints at all (the reference's MemoryConst / MemoryVar only take part in lookups):
        // its columns still enter through the GKR-opening batching term (TOUCH pseudo-instructions, build_chunks)
        SP1HIP_REQUIRE(chips[i].program || chips[i].n_instr == 0, "null constraint program");
        SP1HIP_REQUIRE(chips[i].real_rows <= ((uint64_t)1 << L), "chip taller than 2^max_log_row_count");
        SP1HIP_REQUIRE(chips[i].real_rows == 0 || (chips[i].d_main || chips[i].main_width == 0), "null main trace");
        SP1HIP_REQUIRE(chips[i].real_rows == 0 || (chips[i].d_prep || chips[i].prep_width == 0), "null preprocessed trace");
        total_w += chips[i].main_width + chips[i].prep_width;
    }
    const size_t need = 8 + (size_t)L * (8 + 80) + 16 + 8 + (size_t)L * 16 + 16 + 8 + (size_t)n_chips * 8 + total_w * 16;
    if (!h_proof || *proof_len < need) {
        *proof_len = need;
        set_error("sp1hip_zerocheck_prove: proof buffer too small, need %zu bytes", need);
        return SP1HIP_ERROR_BUFFER_TOO_SMALL;
    }
    hipStream_t s = S(stream);
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    const Ext alpha{{alpha_c.c[0], alpha_c.c[1], alpha_c.c[2], alpha_c.c[3]}};
    const Ext gkr{{gkr_c.c[0], gkr_c.c[1], gkr_c.c[2], gkr_c.c[3]}};
    std::vector<uint32_t> publics(h_publics, h_publics + n_publics);
    DevBuf d_publics;
    SP1HIP_TRY(d_publics.alloc((size_t)n_publics * 4, s));

    int max_constraints = 0;
    for (int i = 0; i < n_chips; i++) max_constraints = std::max<int>(max_constraints, chips[i].num_constraints);
    std::vector<Ext> pows(max_constraints);
    { Ext cur = kb::ext_one(); for (auto& x : pows) { x = cur; cur = cur * alpha; } }

    // Host staging vectors handed to hipMemcpyAsync live until the end of the call (`blob`, the ChipStates, the
    // per-round `keep_*` lists below): no synchronisation is needed just to keep a source buffer valid, and every
    // device->host hand-over goes through the mailbox (round_sync.hpp), so the stream is never drained mid-proof.
    Mailbox mb;
    SP1HIP_TRY(mb.init(s));
    PinnedStage stage;                            // small uploads go through a pinned block (round_sync.hpp)
    SP1HIP_TRY(stage.init(s));
    if (n_publics) SP1HIP_TRY(stage.upload(d_publics.p, publics.data(), (size_t)n_publics * 4));
    std::vector<uint32_t> blob;
    std::vector<std::unique_ptr<ChipState>> st;
    std::vector<Ext> claims;
    size_t oo = 0;
    for (int i = 0; i < n_chips; i++) {
        std::unique_ptr<ChipState> c(new ChipState());
        c->in = &chips[i];
        SP1HIP_TRY(allocate_registers(chips[i].program, chips[i].n_instr, &c->prog, &c->n_regs));
        uint32_t asserts = 0;
        for (uint32_t k = 0; k < chips[i].n_instr; k++) {
            const uint32_t op = chips[i].program[3 * k], a = chips[i].program[3 * k + 1];
            if (op == ZC_ASSERT_ZERO) asserts++;
            if (op == ZC_LOAD_MAIN) SP1HIP_REQUIRE(a < chips[i].main_width, "main column out of range");
            if (op == ZC_LOAD_PREP) SP1HIP_REQUIRE(a < chips[i].prep_width, "preprocessed column out of range");
            if (op == ZC_PUBLIC) SP1HIP_REQUIRE((int)a < n_publics, "public value index out of range");
        }
        SP1HIP_REQUIRE(asserts == chips[i].num_constraints, "num_constraints does not match the program");
        SP1HIP_TRY(build_chunks(chips[i].program, chips[i].n_instr, chips[i].main_width, chips[i].prep_width, ZC_CHUNK_LIMIT,
                                &c->chunks));
        // [alpha^(n-1), ..., alpha, 1] so that the folder matches the verifier's Horner order
        c->alpha_pows.assign(pows.begin(), pows.begin() + chips[i].num_constraints);
        std::reverse(c->alpha_pows.begin(), c->alpha_pows.end());
        { Ext cur = gkr; for (uint32_t k = 0; k < chips[i].main_width + chips[i].prep_width; k++) { c->gkr_pows.push_back(cur); cur = cur * gkr; } }
        c->pad_adj = eval_zero_row(*c, publics.data());
        Ext claim = kb::ext_zero();
        for (uint32_t k = 0; k < chips[i].main_width + chips[i].prep_width; k++, oo++) {
            const Ext o{{h_openings[oo].c[0], h_openings[oo].c[1], h_openings[oo].c[2], h_openings[oo].c[3]}};
            claim = claim + o * c->gkr_pows[k];
        }
        claims.push_back(claim);
        c->rows = chips[i].real_rows;
        c->num_vars = (uint32_t)L;
        c->eq_adj = kb::ext_one();
        c->vgeq = VGeq{(uint32_t)chips[i].real_rows, kb::ext_one(), kb::ext_zero()};
        c->d_main = chips[i].d_main;
        c->d_prep = chips[i].d_prep;
        // programs and power tables of every chip go into ONE blob: one upload for the whole call
        auto pad4 = [&]() { while (blob.size() & 3) blob.push_back(0); };
        pad4();
        c->off_prog = blob.size();
        for (auto& ck : c->chunks) {
            c->chunk_off.push_back((uint32_t)((blob.size() - c->off_prog) / 4));
            blob.insert(blob.end(), ck.prog.begin(), ck.prog.end());
        }
        pad4();
        c->off_alpha = blob.size();
        for (const Ext& e : c->alpha_pows) blob.insert(blob.end(), e.c, e.c + 4);
        c->off_gkr = blob.size();
        for (const Ext& e : c->gkr_pows) blob.insert(blob.end(), e.c, e.c + 4);
        st.push_back(std::move(c));
    }
    DevBuf d_blob;
    SP1HIP_TRY(d_blob.alloc(std::max<size_t>(blob.size(), 4) * 4, s));
    SP1HIP_TRY(stage.upload(d_blob.p, blob.data(), blob.size() * 4));
    for (auto& c : st) {
        c->p_prog = d_blob.u32() + c->off_prog;
        c->p_alpha = d_blob.u32() + c->off_alpha;
        c->p_gkr = d_blob.u32() + c->off_gkr;
    }

    std::vector<Ext> zeta(L);
    memcpy(zeta.data(), h_zeta, (size_t)L * 16);
    const Ext lambda = challenger_sample_ext(challenger);
    DevBuf d_eq;
    SP1HIP_TRY(d_eq.alloc(((size_t)1 << (L - 1)) * 16, s));
    std::vector<UniPoly> msgs;
    std::vector<Ext> point;   // [alpha_last, ..., alpha_first]
    std::vector<Ext> round_claims = claims;
    std::vector<std::array<uint32_t, 16>> sums(n_chips);
    std::vector<uint32_t> h_sums((size_t)n_chips * 16);
    DevBuf d_descs, d_partial, d_sums;       // d_descs: the round's descriptors [ZcDesc.. | ZcChipRange.. | ZcFixDesc..]
    size_t partial_cap = 0, descs_cap = 0;
    std::vector<std::unique_ptr<std::vector<ZcDesc>>> keep_descs;
    std::vector<std::unique_ptr<std::vector<ZcChipRange>>> keep_ranges;
    std::vector<std::unique_ptr<std::vector<ZcFixDesc>>> keep_fds;
    std::vector<std::unique_ptr<std::vector<uint8_t>>> keep_packs;
    SP1HIP_TRY(d_sums.alloc((size_t)n_chips * 64, s));
    for (int r = 0; r < L; r++) {
        const int nv = L - r;                       // variables left
        const Ext last = zeta[nv - 1];
        // eq(zeta[0 .. nv-1), .) is shared by every chip with real rows
        SP1HIP_TRY(sp1hip_partial_lagrange(reinterpret_cast<const sp1hip_ext_t*>(zeta.data()), nv - 1, d_eq.u32(), stream));
        // descriptors: one per (chip with real rows, chunk); a chip's blocks are contiguous
        keep_descs.emplace_back(new std::vector<ZcDesc>());
        keep_ranges.emplace_back(new std::vector<ZcChipRange>());
        std::vector<ZcDesc>& descs = *keep_descs.back();
        std::vector<ZcChipRange>& ranges = *keep_ranges.back();
        std::vector<int> desc_chip;
        uint32_t total_blocks = 0, max_regs = 1, max_instr = 1;
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (c.rows == 0) continue;
            const uint32_t terms = (uint32_t)((c.rows + 1) / 2);
            uint32_t blocks = (terms + 255) / 256;
            if (blocks > 512) blocks = 512;
            ZcChipRange rg{total_blocks, 0, terms - 1, 0};
            for (size_t q = 0; q < c.chunks.size(); q++) {
                ZcDesc d{};
                d.prog = c.p_prog + (size_t)c.chunk_off[q] * 4;
                d.n_instr = (uint32_t)(c.chunks[q].prog.size() / 4);
                d.main = c.d_main; d.prep = c.d_prep; d.main_w = c.in->main_width; d.prep_w = c.in->prep_width;
                d.rows = (uint32_t)c.rows; d.alpha_pows = c.p_alpha; d.gkr_pows = c.p_gkr;
                d.block_start = total_blocks; d.n_blocks = blocks;
                d.alpha_off = c.chunks[q].alpha_off; d.flags = q == 0 ? 1u : 0u;
                total_blocks += blocks;
                max_regs = std::max(max_regs, c.chunks[q].n_regs);
                max_instr = std::max(max_instr, d.n_instr);
                descs.push_back(d);
            }
            rg.n_blocks = total_blocks - rg.block_start;
            ranges.push_back(rg);
            desc_chip.push_back(i);
        }
        const int n_descs = (int)descs.size(), n_ranges = (int)ranges.size();
        // the table update that ends this round needs nothing from the transcript but alpha (a kernel argument): plan
        // it now, so that every descriptor of the round goes up in ONE copy
        keep_fds.emplace_back(new std::vector<ZcFixDesc>());
        std::vector<ZcFixDesc>& fds = *keep_fds.back();
        std::vector<std::unique_ptr<DevBuf>> fresh;
        std::vector<std::pair<int, bool>> owner;   // (chip, is_main)
        uint32_t fix_blocks = 0;
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (c.rows == 0) continue;
            const uint64_t out_rows = (c.rows + 1) / 2;
            for (int which = 0; which < 2; which++) {
                const uint32_t width = which == 0 ? c.in->main_width : c.in->prep_width;
                if (width == 0) continue;
                std::unique_ptr<DevBuf> nb(new DevBuf());
                SP1HIP_TRY(nb->alloc((size_t)out_rows * width * 16, s));
                ZcFixDesc fd{};
                fd.in = which == 0 ? c.d_main : c.d_prep;
                fd.out = nb->u32();
                fd.rows = (uint32_t)c.rows; fd.width = width; fd.block_start = fix_blocks;
                fd.n_blocks = (uint32_t)(((size_t)out_rows * width + 255) / 256);
                fix_blocks += fd.n_blocks;
                fds.push_back(fd);
                fresh.push_back(std::move(nb));
                owner.push_back({i, which == 0});
            }
        }
        const size_t off_ranges = (descs.size() * sizeof(ZcDesc) + 15) & ~(size_t)15;
        const size_t off_fds = (off_ranges + ranges.size() * sizeof(ZcChipRange) + 15) & ~(size_t)15;
        const size_t pack_bytes = off_fds + fds.size() * sizeof(ZcFixDesc);
        keep_packs.emplace_back(new std::vector<uint8_t>(std::max<size_t>(pack_bytes, 16), 0));
        std::vector<uint8_t>& pack = *keep_packs.back();
        if (!descs.empty()) memcpy(pack.data(), descs.data(), descs.size() * sizeof(ZcDesc));
        if (!ranges.empty()) memcpy(pack.data() + off_ranges, ranges.data(), ranges.size() * sizeof(ZcChipRange));
        if (!fds.empty()) memcpy(pack.data() + off_fds, fds.data(), fds.size() * sizeof(ZcFixDesc));
        if (pack.size() > descs_cap) {
            d_descs.release();
            descs_cap = pack.size();
            SP1HIP_TRY(d_descs.alloc(descs_cap, s));
        }
        SP1HIP_TRY(stage.upload(d_descs.p, pack.data(), pack_bytes));
        const ZcChipRange* d_ranges_p = (const ZcChipRange*)((const uint8_t*)d_descs.p + off_ranges);
        const ZcFixDesc* d_fix_p = (const ZcFixDesc*)((const uint8_t*)d_descs.p + off_fds);
        if (n_descs) {
            if ((size_t)total_blocks * 24 * 4 > partial_cap) {
                d_partial.release();
                partial_cap = (size_t)total_blocks * 24 * 4;
                SP1HIP_TRY(d_partial.alloc(partial_cap, s));
            }
            ScopedTimer tm("zerocheck_round", s);      // (the reference's SP1_GPU_ZEROCHECK_ROUND_TIMING switch)
            if (r == 0) SP1HIP_TRY(launch_round<true>(max_regs, (const ZcDesc*)d_descs.p, n_descs, total_blocks, max_instr, d_eq.u32(), 1u << (nv - 1), d_publics.u32(), d_partial.u32(), s));
            else SP1HIP_TRY(launch_round<false>(max_regs, (const ZcDesc*)d_descs.p, n_descs, total_blocks, max_instr, d_eq.u32(), 1u << (nv - 1), d_publics.u32(), d_partial.u32(), s));
            if (r == 0) hipLaunchKernelGGL(zc_reduce_kernel<true>, dim3(n_ranges), dim3(256), 0, s, d_ranges_p, d_partial.u32(), d_eq.u32(), 1u << (nv - 1), d_sums.u32());
            else hipLaunchKernelGGL(zc_reduce_kernel<false>, dim3(n_ranges), dim3(256), 0, s, d_ranges_p, d_partial.u32(), d_eq.u32(), 1u << (nv - 1), d_sums.u32());
            SP1HIP_LAUNCH_CHECK();
            SP1HIP_TRY(mb.fetch(d_sums.p, (size_t)n_ranges * 16, h_sums.data()));
        }
        for (size_t k = 0; k < desc_chip.size(); k++) memcpy(sums[desc_chip[k]].data(), h_sums.data() + k * 16, 64);
        // ---- univariate messages (sum_as_poly.rs:L187-L287)
        // the interpolation nodes {0, 1, 2, 4, b} are the same for every chip of a round: build the five Lagrange
        // basis polynomials once (exact field arithmetic: the result is the reference's interpolation, whatever the
        // operation order) and combine them per chip — 25 extension products instead of a full interpolation each
        const Ext two_c = ext_c(2), four_c = ext_c(4);
        const Ext b_node = (kb::ext_one() - last) * kb::ext_inv(kb::ext_one() - (last + last));
        const std::vector<Ext> nodes{kb::ext_zero(), kb::ext_one(), two_c, four_c, b_node};
        UniPoly basis[5];
        for (int k = 0; k < 5; k++) {
            std::vector<Ext> e(5, kb::ext_zero());
            e[k] = kb::ext_one();
            basis[k] = interpolate(nodes, e);
            basis[k].resize(5, kb::ext_zero());
        }
        std::vector<UniPoly> uni(n_chips);
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (c.rows == 0) { uni[i] = UniPoly(5, kb::ext_zero()); continue; }
            const size_t th = (size_t)((c.rows + 1) / 2) - 1;
            const Ext eq_th{{sums[i][12], sums[i][13], sums[i][14], sums[i][15]}};
            const Ext msb = c.eq_adj * eq_th;
            const Ext y0s{{sums[i][0], sums[i][1], sums[i][2], sums[i][3]}}, y2s{{sums[i][4], sums[i][5], sums[i][6], sums[i][7]}},
                y4s{{sums[i][8], sums[i][9], sums[i][10], sums[i][11]}};
            const Ext two = ext_c(2), four = ext_c(4), three = ext_c(3), seven = ext_c(7);
            const Ext v0 = c.vgeq.fix(kb::ext_zero()).at(th), v2 = c.vgeq.fix(two).at(th), v4 = c.vgeq.fix(four).at(th);
            const Ext f0 = kb::ext_one() - last;
            const Ext y0 = y0s * (f0 * c.eq_adj) - c.pad_adj * v0 * msb * f0;
            const Ext f2 = last * three - kb::ext_one();
            const Ext y2 = y2s * (f2 * c.eq_adj) - c.pad_adj * v2 * msb * f2;
            const Ext f4 = last * seven - three;
            const Ext y4 = y4s * (f4 * c.eq_adj) - c.pad_adj * v4 * msb * f4;
            const Ext ys[4] = {y0, round_claims[i] - y0, y2, y4};          // the fifth value, at b, is zero
            uni[i].assign(5, kb::ext_zero());
            for (int d = 0; d < 5; d++)
                for (int k = 0; k < 4; k++) uni[i][d] = uni[i][d] + ys[k] * basis[k][d];
        }
        UniPoly rlc{kb::ext_zero()};
        for (auto& u : uni) rlc = uni_add(uni_scale(rlc, lambda), u);
        for (auto& cf : rlc)
            for (int k = 0; k < 4; k++) challenger_observe(challenger, cf.c[k]);
        msgs.push_back(rlc);
        const Ext a_r = challenger_sample_ext(challenger);
        point.insert(point.begin(), a_r);
        for (int i = 0; i < n_chips; i++) {
            round_claims[i] = uni_eval(uni[i], a_r);
            st[i]->uni = uni[i];
        }
        // ---- fix the last variable of every table (fix_last_variable.rs): one launch for all chips (planned above)
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            c.vgeq = c.vgeq.fix(a_r);
            if (c.rows == 0) continue;
            c.eq_adj = c.eq_adj * (a_r * last + (kb::ext_one() - a_r) * (kb::ext_one() - last));
        }
        if (!fds.empty()) {
            ScopedTimer tm("zerocheck_fix", s);
            if (r == 0) hipLaunchKernelGGL(zc_fix_kernel<true>, dim3(fix_blocks), dim3(256), 0, s, d_fix_p, (int)fds.size(), a_r);
            else hipLaunchKernelGGL(zc_fix_kernel<false>, dim3(fix_blocks), dim3(256), 0, s, d_fix_p, (int)fds.size(), a_r);
            SP1HIP_LAUNCH_CHECK();
            for (size_t k = 0; k < fds.size(); k++) {     // the arena is stream-ordered: the old table is recycled behind this launch
                ChipState& c = *st[owner[k].first];
                if (owner[k].second) { c.main_buf = std::move(fresh[k]); c.d_main = c.main_buf->u32(); }   // old table released
                else { c.prep_buf = std::move(fresh[k]); c.d_prep = c.prep_buf->u32(); }
            }
        }
        for (int i = 0; i < n_chips; i++)
            if (st[i]->rows) st[i]->rows = (st[i]->rows + 1) / 2;
    }
    // ---- proof: PartialSumcheckProof + per-chip component evaluations (prep then main)
    ByteOut w;
    w.u64((uint64_t)L);
    for (auto& m : msgs) { w.u64(m.size()); for (auto& cf : m) w.ext(cf); }
    Ext claimed = kb::ext_zero(), final_eval = kb::ext_zero();
    for (auto& cl : claims) claimed = claimed * lambda + cl;
    for (int i = 0; i < n_chips; i++) final_eval = final_eval * lambda + uni_eval(st[i]->uni, point.front());
    w.ext(claimed);
    w.u64(point.size());
    for (auto& x : point) w.ext(x);
    w.ext(final_eval);
    w.u64((uint64_t)n_chips);
    std::vector<std::vector<Ext>> chip_evals(n_chips);
    {   // one row is left of every table: ext [1 x w] = w*4 words (col, coord). Gather them all, one hand-over.
        std::vector<ZcGatherDesc> gd;
        size_t total_words = 0;
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            const uint32_t wp = c.in->prep_width, wm = c.in->main_width;
            if (c.rows && wp) gd.push_back({c.d_prep, wp * 4, (uint32_t)total_words});
            total_words += (size_t)wp * 4;
            if (c.rows && wm) gd.push_back({c.d_main, wm * 4, (uint32_t)total_words});
            total_words += (size_t)wm * 4;
        }
        std::vector<uint32_t> flat(total_words, 0);
        if (!gd.empty()) {
            DevBuf d_gd, d_flat;
            SP1HIP_TRY(d_gd.alloc(gd.size() * sizeof(ZcGatherDesc), s));
            SP1HIP_TRY(d_flat.alloc(total_words * 4, s));
            SP1HIP_HIP(hipMemsetAsync(d_flat.p, 0, total_words * 4, s));
            SP1HIP_TRY(stage.upload(d_gd.p, gd.data(), gd.size() * sizeof(ZcGatherDesc)));
            hipLaunchKernelGGL(zc_gather_kernel, dim3((unsigned)gd.size()), dim3(256), 0, s, (const ZcGatherDesc*)d_gd.p, d_flat.u32());
            SP1HIP_LAUNCH_CHECK();
            SP1HIP_TRY(mb.fetch(d_flat.p, total_words, flat.data()));     // also keeps `gd` valid long enough
        }
        size_t off = 0;
        for (int i = 0; i < n_chips; i++) {
            const uint32_t wtot = st[i]->in->prep_width + st[i]->in->main_width;
            for (uint32_t k = 0; k < wtot; k++, off += 4)
                chip_evals[i].push_back(Ext{{flat[off], flat[off + 1], flat[off + 2], flat[off + 3]}});
            w.u64(chip_evals[i].size());
            for (auto& e : chip_evals[i]) w.ext(e);
        }
    }
    // observe the openings (shard.rs:L609-L640)
    challenger_observe(challenger, kb::to_monty((uint32_t)n_chips));
    for (int i = 0; i < n_chips; i++) {
        const uint32_t wp = st[i]->in->prep_width, wm = st[i]->in->main_width;
        challenger_observe(challenger, kb::to_monty(wp));
        for (uint32_t k = 0; k < wp; k++) for (int q = 0; q < 4; q++) challenger_observe(challenger, chip_evals[i][k].c[q]);
        challenger_observe(challenger, kb::to_monty(wm));
        for (uint32_t k = 0; k < wm; k++) for (int q = 0; q < 4; q++) challenger_observe(challenger, chip_evals[i][wp + k].c[q]);
    }
    if (w.b.size() != need) { set_error("internal error: zerocheck proof size %zu != %zu", w.b.size(), need); return SP1HIP_ERROR_RUNTIME; }
    memcpy(h_proof, w.b.data(), need);
    *proof_len = need;
    return SP1HIP_SUCCESS;
}

namespace sp1hip {
// standalone form of the per-round table update, with the reference's per-column padding value
template <bool FIRST>
__global__ __launch_bounds__(256) void fix_last_variable_kernel(const uint32_t* __restrict__ in, uint32_t rows, uint32_t width,
                                                                kb::Ext alpha, const uint32_t* __restrict__ padding,
                                                                uint32_t* __restrict__ out) {
    using K = KT<FIRST>;
    const uint32_t out_rows = (rows + 1) / 2;
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= (size_t)out_rows * width) return;
    const uint32_t c = (uint32_t)(t / out_rows), i = (uint32_t)(t % out_rows);
    typename K::T x = K::load(in, c, rows, 2 * i), y;
    if (2 * i + 1 < rows) y = K::load(in, c, rows, 2 * i + 1);
    else if (!padding) y = K::zero();
    else y = K::load(padding, c, 1, 0);                   // a one-row table in the same layout
    const kb::Ext r = kb::ext_add(K::scale(alpha, K::sub(y, x)), K::to_ext(x));
#pragma unroll
    for (int q = 0; q < 4; q++) out[((size_t)c * 4 + q) * out_rows + i] = r.c[q];
}
}  // namespace sp1hip

extern "C" int sp1hip_fix_last_variable(const uint32_t* d_in, uint64_t rows, uint32_t width, int in_is_ext, sp1hip_ext_t alpha,
                                        const uint32_t* d_padding, uint32_t* d_out, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(rows < ((uint64_t)1 << 32), "too many rows");
    if (rows == 0 || width == 0) return SP1HIP_SUCCESS;
    SP1HIP_REQUIRE(d_in && d_out && d_in != d_out, "bad buffers");
    const kb::Ext a{{alpha.c[0], alpha.c[1], alpha.c[2], alpha.c[3]}};
    const uint64_t total = ((rows + 1) / 2) * width;
    SP1HIP_REQUIRE((total + 255) / 256 < ((uint64_t)1 << 31), "table too large for one launch");
    const dim3 grid((unsigned)((total + 255) / 256));
    if (in_is_ext) hipLaunchKernelGGL(fix_last_variable_kernel<false>, grid, dim3(256), 0, S(stream), d_in, (uint32_t)rows, width, a, d_padding, d_out);
    else hipLaunchKernelGGL(fix_last_variable_kernel<true>, grid, dim3(256), 0, S(stream), d_in, (uint32_t)rows, width, a, d_padding, d_out);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

extern "C" int sp1hip_zerocheck_prove(const sp1hip_zc_chip_t* chips, int n_chips, int max_log_row_count,
                                      const sp1hip_ext_t* h_zeta, const sp1hip_ext_t* h_openings, sp1hip_ext_t alpha,
                                      sp1hip_ext_t gkr_batch, const uint32_t* h_publics, int n_publics,
                                      sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                                      sp1hip_stream_t stream) {
    // the caller's transcript only advances if the proof is produced
    sp1hip_challenger_t* backup = nullptr;
    if (challenger) SP1HIP_TRY(sp1hip_challenger_clone(challenger, &backup));
    const int st = zerocheck_prove_impl(chips, n_chips, max_log_row_count, h_zeta, h_openings, alpha, gkr_batch, h_publics,
                                        n_publics, challenger, h_proof, proof_len, stream);
    if (st != SP1HIP_SUCCESS && challenger) challenger_restore(challenger, backup);
    sp1hip_challenger_free(backup);
    return st;
}
