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
// per column). The register file lives in LDS (up to 64 extension registers per lane; beyond that, VGPR / scratch
// tiers); hinted sub-AIRs (Poseidon2 permutation, septic curve, Keccak-f round) are evaluated by fused pieces with
// their state in VGPRs instead (zc_poseidon2.hpp). Per-chip compiled kernels were built in round 3,
// measured slower than this interpreter and removed in round 4.
#include <algorithm>
#include <array>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "device_ctx.hpp"
#include "round_sync.hpp"
#include "zc_device.hpp"
#include "zc_poseidon2.hpp"
#include "zc_keccak.hpp"
#include "zc_mul.hpp"
#include "zc_poly.hpp"

namespace sp1hip {

// ---- register file -------------------------------------------------------------------------------
// The program is wave-uniform, so register numbers are SGPR values. The file lives in LDS (the only tier since round 5: the
// VGPR-vector files of rounds 1-4 — s_set_gpr_idx triples per word — and the per-lane scratch files for programs with more
// than 64 live values — 4 to 16 KB of scratch per lane — are gone; a program whose register file does not fit the 160 KB of
// LDS even for one wave is cut into finer chunks by the planner, zc_wg_for / plan_round).
template <bool FIRST, int MAXR> struct RegFile;

// The file: register i of a lane at slot i * (workgroup width) + lane (16 B slots for extension values: one
// ds_read_b128 / ds_write_b128 per access, conflict-free). Indexing a VGPR vector with a wave-uniform index costs an
// s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off triple per word — ~36 instructions of pure register traffic around a
// 12-instruction extension add; the LDS file makes an interpreted op cost its arithmetic plus three LDS accesses,
// and leaves the VGPRs to the arithmetic (measured per-op cost: add 75 -> ~20 instructions, multiply 147 -> ~100).
// The slot stride is the workgroup size: programs with many live values run in narrower workgroups (128 / 64 lanes) so
// that the file still fits the LDS budget (launch_round).
// (pointers carry the LDS address space explicitly: a generic pointer here turns every access into a FLAT instruction)
typedef uint32_t zc_lds_word_t __attribute__((address_space(3)));
typedef uint32_t zc_lds_quad_t __attribute__((ext_vector_type(4), address_space(3)));
typedef uint32_t zc_quad_t __attribute__((ext_vector_type(4)));
template <> struct RegFile<true, 0> {
    zc_lds_word_t* base;           // this lane's slot of register 0
    uint32_t stride;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return base[i * stride]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) { base[i * stride] = v; }
};
template <> struct RegFile<false, 0> {
    zc_lds_quad_t* base;
    uint32_t stride;
    __device__ __forceinline__ kb::Ext get(uint32_t i) const { const zc_quad_t v = base[i * stride]; return kb::Ext{{v.x, v.y, v.z, v.w}}; }
    __device__ __forceinline__ void set(uint32_t i, const kb::Ext& v) { zc_quad_t q = {v.c[0], v.c[1], v.c[2], v.c[3]}; base[i * stride] = q; }
};

constexpr uint32_t ZC_GKR_FLAG = 0x100u;   // bits 8..11: column j of this load is the first load of that column (host)
constexpr uint32_t ZC_A_PREV = 0x1000u, ZC_B_PREV = 0x2000u;   // operand = the value the previous instruction produced
constexpr uint32_t ZC_DST_TEMP = 0x4000u;  // the result is only forwarded, never stored in the register file
                                           // bits 16..17: (number of consecutive columns of a LOAD) - 1
constexpr uint32_t ZC_TOUCH = 9;           // pseudo-op: column never loaded by the constraints (GKR term only)
// internal forms with an immediate operand (host peephole `fold_immediates`: an ADD / SUB / MUL one of whose operands is
// a CONST): the constant travels in the instruction word — no register, no LDS access, and in the extension rounds a
// constant factor is 4 base products instead of a full 16-product extension multiply
constexpr uint32_t ZC_ADDC = 10, ZC_SUBC = 11, ZC_CSUB = 12, ZC_MULC = 13;
// fused multiply-add by a constant (register allocation, host): acc + term * c for a MULC whose only use is the ADD / SUB
// (as subtrahend, c negated) that follows it — the linear combinations real chips are full of (limb recompositions, the
// closed-form Poseidon2 rounds) become ONE dispatch per term, and the running sum is forwarded from instruction to
// instruction in VGPRs instead of going through the register file. Word: op | flags, dst | (acc register << 16), term
// register, c; ZC_A_PREV: term = previous value, ZC_B_PREV: acc = previous value.
constexpr uint32_t ZC_MADC = 14;
constexpr uint32_t ZC_RSUB = 15;           // b - a: a SUB whose forwarded operand is the subtrahend (register allocation, host)
// acc + a * b / acc - a * b: a MUL whose single use is the ADD / SUB that is the next value-producing instruction (host peephole in
// allocate_registers, like MADC). The sums of products real chips are made of (MulOperation: 136 byte products in 16 chains) are
// emitted term by term with the running sum as the previous value, so a chain becomes MUL, MAD, MAD, ... with the sum FORWARDED:
// two LDS reads per term and no write, instead of a MUL (two reads) and an ADD (a read and a write) — and half the decodes.
// Word: op | flags, dst | (acc register << 16), a, b; ZC_B_PREV: acc = previous value (a, b from the file); else ZC_A_PREV: a =
// previous value.
constexpr uint32_t ZC_MAD = 16, ZC_MSB = 17;
constexpr uint32_t ZC_MONO_MIN_TERMS = 1024; // rounds with at least this many row pairs run a chip's program in ONE piece
constexpr uint32_t ZC_CHUNK_LIMIT = 96;    // target instructions per chunk (host-side program splitting)
constexpr uint32_t ZC_CHUNK_HARD_MAX = 320; // a chunk may grow to this while its asserts share most of their cones
// the last rounds (at most ZC_FINE_MAX_TERMS row pairs per chip: one wave, mostly idle lanes) are pure latency — one wave
// interprets a chunk serially at ~0.35 us per instruction — so they run a third form of the program, cut into pieces of
// ~ZC_FINE_LIMIT instructions with no regard for recomputation: more workgroups, each a third as long
constexpr uint32_t ZC_FINE_LIMIT = 32, ZC_FINE_MAX_TERMS = 64;
// rounds with at most this many workgroups (x 3 nodes) launch their groups / fused pieces on fork streams; the default forks every round
// (the large rounds gain the overlap of one launch's tail with the next one's head). SP1HIP_ZC_FORK_MAX_BLOCKS overrides (A/B runs).
constexpr uint32_t ZC_FORK_MAX_BLOCKS = 1u << 30;
// rounds with at most this many workgroups are "small": every launch is at its latency floor (the two septic kinds then share one launch)
constexpr uint32_t ZC_SMALL_ROUND_WGS = 16384;

typedef uint32_t zc_word_t __attribute__((ext_vector_type(4)));       // one instruction: op | flags, dst, a, b
typedef const zc_word_t __attribute__((address_space(4)))* zc_const_prog_t;

// ---- the first two rounds in one pass over the base-field traces ("bivariate", the reference's
// sp1-gpu/crates/sys/include/zerocheck/bivariate.cuh:L1-L118 restated for this interpreter) ---------------------------------
// Rows are taken four at a time (row 4 q + 2 X + Y: Y is the last variable, bound by round 0, X the one round 1 binds) and the
// constraint polynomial is summed on the grid {0, 1, 2, 4}^2 minus its four boolean corners (constraints vanish on real rows,
// and a padded row's value cancels against the geq correction): per node e the kernels leave
//     A_e = sum_q eq(q) C(T_q(X_e, Y_e)),      T_q(X, Y) = r00 + X (r10 - r00) + Y (r01 - r00) + X Y (r11 - r10 - r01 + r00)
// with eq over the nv - 2 variables of the quad index, and the four corner sums B of the (linear) GKR batching term; the host
// assembles BOTH round messages from them (zerocheck_prove_impl): round 0 needs H(X, t) for X in {0, 1}, t in {0, 2, 4}, round 1
// the cubic through H(t, 0), H(t, 1), H(t, 2), H(t, 4) at the first challenge. Everything is base-field arithmetic — round 1's
// extension-field pass over the once-folded tables (the most expensive round of the sequential form) and one of the two table
// updates disappear. Node order (X, Y): (0,2) (0,4) (1,2) (1,4) (2,0) (2,1) (2,2) (2,4) (4,0) (4,1) (4,2) (4,4).
constexpr int ZC_BIV_NODES = 12;
struct ZcBivNode { uint32_t cx, cy, cxy; };
__host__ __device__ __forceinline__ ZcBivNode zc_biv_node(uint32_t e) {       // wave-uniform e: the fields stay in SGPRs
    constexpr uint32_t XS[12] = {0, 0, 1, 1, 2, 2, 2, 2, 4, 4, 4, 4}, YS[12] = {2, 4, 2, 4, 0, 1, 2, 4, 0, 1, 2, 4};
    return ZcBivNode{XS[e], YS[e], XS[e] * YS[e]};
}
// v < 2^36 -> v mod p, reduced: with v = t 2^31 + lo, v - t p = lo + t (2^24 - 1) < 2 p
__host__ __device__ __forceinline__ uint32_t zc_reduce36(uint64_t v) {
    const uint32_t t = (uint32_t)(v >> 31), lo = (uint32_t)v & 0x7fffffffu;
    const uint32_t r = lo + t * 0xffffffu;
    return kb::umin(r, r - kb::P);
}
__host__ __device__ __forceinline__ uint32_t zc_biv_interp(uint32_t r00, uint32_t r01, uint32_t r10, uint32_t r11, const ZcBivNode& nd) {
    const uint32_t dy = kb::sub(r01, r00), dx = kb::sub(r10, r00), dxy = kb::sub(kb::sub(r11, r10), dy);
    return zc_reduce36((uint64_t)r00 + (uint64_t)nd.cx * dx + (uint64_t)nd.cy * dy + (uint64_t)nd.cxy * dxy);   // <= (1 + 4 + 4 + 16) p
}
// column `col` of the quad q at node nd (rows past the table's height are zero: the virtual padding)
__device__ __forceinline__ uint32_t zc_biv_leaf(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t q, const ZcBivNode& nd) {
    const zc_global_words_t g = (zc_global_words_t)tbl + (size_t)col * rows;
    const uint32_t r = 4 * q;
    const uint32_t r00 = g[r], r01 = r + 1 < rows ? g[r + 1] : 0u, r10 = r + 2 < rows ? g[r + 2] : 0u, r11 = r + 3 < rows ? g[r + 3] : 0u;
    return zc_biv_interp(r00, r01, r10, r11, nd);
}

// Four nodes per pass: the interpreter's cost per base-field operation is mostly decode and register-file traffic, so one pass
// of the program carries the values of FOUR grid nodes (nodes 4 g .. 4 g + 3) in an Ext-shaped container — the instruction is
// decoded once, the register file is the extension rounds' (16-byte slots), the arithmetic is element-wise.
struct KT4 {
    using T = kb::Ext;
    static __device__ __forceinline__ T zero() { return kb::ext_zero(); }
    static __device__ __forceinline__ T from_f(uint32_t x) { return kb::Ext{{x, x, x, x}}; }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return kb::ext_add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return kb::ext_sub(a, b); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) {
        return kb::Ext{{kb::mul(a.c[0], b.c[0]), kb::mul(a.c[1], b.c[1]), kb::mul(a.c[2], b.c[2]), kb::mul(a.c[3], b.c[3])}};
    }
};
struct KC4 {
    static __device__ __forceinline__ kb::Ext addc(const kb::Ext& a, uint32_t c) { return kb::Ext{{kb::add(a.c[0], c), kb::add(a.c[1], c), kb::add(a.c[2], c), kb::add(a.c[3], c)}}; }
    static __device__ __forceinline__ kb::Ext subc(const kb::Ext& a, uint32_t c) { return kb::Ext{{kb::sub(a.c[0], c), kb::sub(a.c[1], c), kb::sub(a.c[2], c), kb::sub(a.c[3], c)}}; }
    static __device__ __forceinline__ kb::Ext csub(uint32_t c, const kb::Ext& a) { return kb::Ext{{kb::sub(c, a.c[0]), kb::sub(c, a.c[1]), kb::sub(c, a.c[2]), kb::sub(c, a.c[3])}}; }
    static __device__ __forceinline__ kb::Ext mulc(const kb::Ext& a, uint32_t c) { return kb::ext_mul_base(a, c); }
};
// column `col` of the quad q at the four nodes of group g: the rows are loaded once
__device__ __forceinline__ kb::Ext zc_biv_leaf4(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t q, uint32_t grp) {
    const zc_global_words_t g = (zc_global_words_t)tbl + (size_t)col * rows;
    const uint32_t r = 4 * q;
    const uint32_t r00 = g[r], r01 = r + 1 < rows ? g[r + 1] : 0u, r10 = r + 2 < rows ? g[r + 2] : 0u, r11 = r + 3 < rows ? g[r + 3] : 0u;
    const uint32_t dy = kb::sub(r01, r00), dx = kb::sub(r10, r00), dxy = kb::sub(kb::sub(r11, r10), dy);
    kb::Ext out;
#pragma unroll
    for (uint32_t n = 0; n < 4; n++) {
        const ZcBivNode nd = zc_biv_node(4 * grp + n);
        out.c[n] = zc_reduce36((uint64_t)r00 + (uint64_t)nd.cx * dx + (uint64_t)nd.cy * dy + (uint64_t)nd.cxy * dxy);
    }
    return out;
}

// One pass of the program at node t. With `gkr`, the first load of every column also accumulates
// gkr_pow[column] * value into *g (main columns first, then preprocessed): the batching term costs no
// extra loads. `prog` points to LDS (or global memory for very long programs).
// BIV: i is a quad index and t a node GROUP of the bivariate grid (nodes 4 t .. 4 t + 3, KT4: four base-field values per
// register, the extension rounds' register file); the four constraint sums are ADDED to g[0..4), no GKR term here.
template <bool FIRST, int MAXR, bool BIV = false, typename PROG>
__device__ __forceinline__ kb::Ext run_program(RegFile<(BIV ? false : FIRST), MAXR>& reg, PROG prog, const ZcDesc& d,
                                               const uint32_t* __restrict__ publics, uint32_t i, int t, const bool gkr, kb::Ext* g) {
    using K = typename std::conditional<BIV, KT4, KT<FIRST>>::type;
    using KCc = typename std::conditional<BIV, KC4, KC<FIRST>>::type;
    using T = typename K::T;
    kb::Ext acc = kb::ext_zero();
    T prev = K::zero();                   // the value the last value-producing instruction produced (operand forwarding)
    auto next = prog[0];                  // instruction words are fetched one instruction ahead of their use
    for (uint32_t k = 0; k < d.n_instr; k++) {
        const auto w = next;              // wave-uniform: decode once, keep the fields in SGPRs
        if (k + 1 < d.n_instr) next = prog[k + 1];
        const uint32_t opw = __builtin_amdgcn_readfirstlane(w.x), dst = __builtin_amdgcn_readfirstlane(w.y);
        const uint32_t x = __builtin_amdgcn_readfirstlane(w.z), y = __builtin_amdgcn_readfirstlane(w.w);
        const uint32_t op = opw & 0xffu;
        if (op <= ZC_LOAD_PREP) {         // 1..4 consecutive columns: every global load is in flight before the first use
            const uint32_t cnt = ((opw >> 16) & 3u) + 1;
            const uint32_t* tbl = op == ZC_LOAD_MAIN ? d.main : d.prep;
            const uint32_t gbase = op == ZC_LOAD_MAIN ? 0u : d.main_w;
            if constexpr (BIV) {
                T v[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (j < cnt) v[j] = zc_biv_leaf4(tbl, x + j, d.rows, i, (uint32_t)t);
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (j < cnt) {
                        if (!(opw & ZC_DST_TEMP)) reg.set(dst + j, v[j]);
                        prev = v[j];
                    }
                continue;
            } else {
            T r0[4], r1[4];
            const bool odd = 2 * i + 1 < d.rows;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (j < cnt) {
                    r0[j] = K::load(tbl, x + j, d.rows, 2 * i);
                    r1[j] = (t != 0 && odd) ? K::load(tbl, x + j, d.rows, 2 * i + 1) : K::zero();
                }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (j < cnt) {
                    T v = r0[j];
                    if (t != 0) {
                        const T s2 = K::add(K::sub(r1[j], r0[j]), K::sub(r1[j], r0[j]));
                        v = t == 2 ? K::add(s2, r0[j]) : K::add(K::add(s2, s2), r0[j]);
                    }
                    if (gkr && (opw & (ZC_GKR_FLAG << j))) *g = kb::ext_add(*g, K::scale(load_ext_aos(d.gkr_pows, gbase + x + j), v));
                    if (!(opw & ZC_DST_TEMP)) reg.set(dst + j, v);
                    prev = v;
                }
            continue;
            }
        }
        if (op == ZC_TOUCH) {
            if constexpr (!BIV) {
                if (gkr) {
                    T v = leaf<FIRST>(y ? d.prep : d.main, x, d.rows, i, t);
                    *g = kb::ext_add(*g, K::scale(load_ext_aos(d.gkr_pows, (y ? d.main_w : 0u) + x), v));
                }
            }
            continue;
        }
        if (op == ZC_ASSERT_ZERO) {                                   // y: the constraint's index; `prev` stays what it was
            const T a = (opw & ZC_A_PREV) ? prev : reg.get(x);
            if constexpr (BIV) {
                const kb::Ext pw = load_ext_aos(d.alpha_pows, y);
#pragma unroll
                for (int n = 0; n < 4; n++) g[n] = kb::ext_add(g[n], kb::ext_mul_base(pw, a.c[n]));
            } else {
                acc = kb::ext_add(acc, K::scale(load_ext_aos(d.alpha_pows, y), a));
            }
            continue;
        }
        // The forwarded value `prev` is dead once this instruction has read it, so the A operand is loaded INTO it when it
        // is not the forwarded value itself: no operand copies at the merge of the "forwarded" and "register file" paths
        // (they were 8 v_mov per extension-field instruction). The host puts the forwarded operand of a binary
        // instruction first (ADD / MUL commute, SUB becomes RSUB); ZC_B_PREV then means "b is the same value as a".
        T res;
        if (op == ZC_MADC) {
            if (opw & ZC_B_PREV) {                                    // the running sum is the forwarded value
                const T term = (opw & ZC_A_PREV) ? prev : reg.get(x);
                res = K::add(prev, KCc::mulc(term, y));
            } else {
                const T accv = reg.get(dst >> 16);
                if (!(opw & ZC_A_PREV)) prev = reg.get(x);
                res = K::add(accv, KCc::mulc(prev, y));
            }
        } else if (op == ZC_MAD || op == ZC_MSB) {
            T accv;
            if (opw & ZC_B_PREV) {                                    // the running sum is the forwarded value
                accv = prev;
                prev = reg.get(x);
            } else {
                accv = reg.get(dst >> 16);
                if (!(opw & ZC_A_PREV)) prev = reg.get(x);
            }
            const T m = K::mul(prev, reg.get(y));
            res = op == ZC_MAD ? K::add(accv, m) : K::sub(accv, m);
        } else if (op == ZC_CONST) {
            res = K::from_f(x);                                       // host pre-converts to Montgomery
        } else if (op == ZC_PUBLIC) {
            res = K::from_f(publics[x]);
        } else {
            if (!(opw & ZC_A_PREV)) prev = reg.get(x);
            switch (op) {
                case ZC_ADD: res = K::add(prev, (opw & ZC_B_PREV) ? prev : reg.get(y)); break;
                case ZC_SUB: res = K::sub(prev, (opw & ZC_B_PREV) ? prev : reg.get(y)); break;
                case ZC_RSUB: res = K::sub(reg.get(y), prev); break;
                case ZC_MUL: res = K::mul(prev, (opw & ZC_B_PREV) ? prev : reg.get(y)); break;
                case ZC_NEG: res = K::sub(K::zero(), prev); break;
                case ZC_ADDC: res = KCc::addc(prev, y); break;
                case ZC_SUBC: res = KCc::subc(prev, y); break;
                case ZC_CSUB: res = KCc::csub(y, prev); break;
                default: res = KCc::mulc(prev, y); break;      // ZC_MULC
            }
        }
        prev = res;
        if (!(opw & ZC_DST_TEMP)) reg.set(dst & 0xffffu, res);
    }
    return acc;
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

// the sums of one range from its partials -> (y0, y2, y4, eq[th]); all threads of the workgroup call this, the results are
// valid in threads 0..3 (word k of each extension element)
template <bool FIRST>
__device__ __forceinline__ void zc_reduce_range(const ZcChipRange& d, const uint32_t* __restrict__ partial, const uint32_t* __restrict__ eq,
                                                uint32_t eq_len, uint32_t (&acc)[10][24], uint32_t& y0, uint32_t& y2, uint32_t& y4, uint32_t& e) {
    const uint32_t word = threadIdx.x % 24, grp = threadIdx.x / 24, n_grp = min(blockDim.x / 24u, 10u);
    auto ld = [&](const uint32_t* q) -> uint32_t { return *q; };
    if (grp < n_grp) {
        // eight independent partial sums: the loads of a lane are then eight deep in flight instead of one behind each add
        // (a tall chip has thousands of blocks: the plain loop was 100-130 us in each of the first three rounds)
        const uint32_t* p = partial + (size_t)d.block_start * 24 + word;
        uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t b = grp;
        for (; b + 7 * n_grp < d.n_blocks; b += 8 * n_grp)
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = kb::add(a[u], ld(p + (size_t)(b + n_grp * u) * 24));
        for (; b < d.n_blocks; b += n_grp) a[0] = kb::add(a[0], ld(p + (size_t)b * 24));
        acc[grp][word] = kb::add(kb::add(kb::add(a[0], a[1]), kb::add(a[2], a[3])), kb::add(kb::add(a[4], a[5]), kb::add(a[6], a[7])));
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        uint32_t a = 0;
        for (uint32_t g = 0; g < n_grp; g++) a = kb::add(a, acc[g][threadIdx.x]);
        acc[0][threadIdx.x] = a;
    }
    __syncthreads();
    y0 = y2 = y4 = e = 0;
    if (threadIdx.x < 4) {
        const uint32_t k = threadIdx.x;
        // S[p][0..4) = A of pass p, S[p][4..8) = B of pass p
        const uint32_t A0 = acc[0][k], B0 = acc[0][4 + k], A1 = acc[0][8 + k], B1 = acc[0][12 + k], A2 = acc[0][16 + k];
        if (FIRST) {       // g0 = A0, g2 = B0, C(2) = A1, C(4) = A2
            y0 = A0;
            y2 = kb::add(A1, B0);
            y4 = kb::add(A2, kb::sub(kb::add(B0, B0), A0));
        } else {           // C(0) = A0, g0 = B0, C(2) = A1, g2 = B1, C(4) = A2
            y0 = kb::add(A0, B0);
            y2 = kb::add(A1, B1);
            y4 = kb::add(A2, kb::sub(kb::add(B1, B1), B0));
        }
        e = d.th < eq_len ? eq[(size_t)k * eq_len + d.th] : 0u;
    }
}
// payload words [1 + 16 range ..) of the host slot: system-scope stores from threads 0..3
__device__ __forceinline__ void zc_store_host_sums(volatile uint32_t* host_slot, uint32_t range, uint32_t y0, uint32_t y2, uint32_t y4, uint32_t e) {
    if (threadIdx.x < 4) {
        const uint32_t k = threadIdx.x;
        uint32_t* h = const_cast<uint32_t*>(host_slot) + 1 + (size_t)range * 16;
        __hip_atomic_store(h + k, y0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(h + 4 + k, y2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(h + 8 + k, y4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(h + 12 + k, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One launch per sumcheck round covers EVERY chip and the three interpolation nodes:
//   blockIdx.x = 3 b + p -> (chip, block b of 256 row pairs), pass p (node t = 2p).
// A pass-p workgroup writes two extension partial sums [A | B] (8 words):
//   round 0 :  p=0: A = sum eq g(0), B = sum eq g(2)   (GKR batching term only; constraints vanish at 0)
//              p=1: A = sum eq C(2)                     p=2: A = sum eq C(4)
//   later   :  p=0: A = sum eq C(0), B = sum eq g(0)    p=1: A = sum eq C(2), B = sum eq g(2)    p=2: A = sum eq C(4)
// g(4) = 2 g(2) - g(0) is linear, so the three nodes can run in different workgroups and the late, tiny
// rounds (latency-bound: one wave interprets the whole program serially) run all chips and nodes at once.
// STAGED: the program is copied to LDS once per workgroup (short chunked programs); otherwise every wave streams it
// from global memory through the scalar cache (constant address space: wave-uniform s_load_dwordx4), which leaves the
// whole LDS budget to the register file — the form used for a chip's undivided program in the large rounds.

template <bool FIRST, int MAXR, bool STAGED>
__global__ __launch_bounds__(256) void zc_round_kernel(const ZcDesc* __restrict__ descs, int n_descs,
                                                       const uint32_t* __restrict__ eq, uint32_t eq_len,
                                                       const uint32_t* __restrict__ publics, uint32_t* __restrict__ partial,
                                                       uint32_t rf_off, uint32_t block_base, uint32_t fused_flag) {
    using K = KT<FIRST>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* red = lds;                                   // [4][8] reduction scratch
    uint4* lprog = reinterpret_cast<uint4*>(lds + 32);
    RegFile<FIRST, MAXR> reg;
    if constexpr (MAXR == 0) {                             // LDS file behind the (staged) program
        reg.base = (decltype(reg.base))(lds + rf_off) + threadIdx.x;
        reg.stride = blockDim.x;
    }
    if (threadIdx.x < 32) red[threadIdx.x] = 0;            // workgroups narrower than 4 waves leave slots untouched
    // The three nodes of a block of row pairs are three CONSECUTIVE workgroups: dispatched together (to different XCDs),
    // their re-reads of the same table rows meet in the memory-side cache instead of HBM. `fused` (A/B knob
    // SP1HIP_ZC_FUSE_NODES=1) makes one workgroup evaluate all three nodes instead; measured slower, see the host side.
    const bool fused = (fused_flag != 0);
    const uint32_t bid = block_base + (fused ? blockIdx.x : blockIdx.x / 3u);
    const int only_pass = fused ? -1 : (int)(blockIdx.x % 3u);
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    if constexpr (STAGED) {
        const uint4* src = reinterpret_cast<const uint4*>(d.prog);
        for (uint32_t k = threadIdx.x; k < d.n_instr; k += blockDim.x) lprog[k] = src[k];
    }
    __syncthreads();
    const uint32_t terms = (d.rows + 1) / 2;
    kb::Ext sa[3], sb[3];
#pragma unroll
    for (int p = 0; p < 3; p++) { sa[p] = kb::ext_zero(); sb[p] = kb::ext_zero(); }
    // a block is d.block_pairs row pairs (= the workgroup width of the chip's launch group: one pass of the program
    // per workgroup while the round is large; the partial-sum layout only knows blocks)
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < terms; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, terms); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
#pragma unroll
        for (int pass = 0; pass < 3; pass++) {
            if (only_pass >= 0 && pass != only_pass) continue;
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
            } else if constexpr (STAGED) {
                va = run_program<FIRST, MAXR>(reg, (const zc_word_t*)lprog, d, publics, i, 2 * pass, !FIRST && pass < 2, &vb);
            } else {
                va = run_program<FIRST, MAXR>(reg, (zc_const_prog_t)(uintptr_t)d.prog, d, publics, i, 2 * pass, !FIRST && pass < 2, &vb);
            }
            sa[pass] = kb::ext_add(sa[pass], kb::ext_mul(va, e));
            sb[pass] = kb::ext_add(sb[pass], kb::ext_mul(vb, e));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
        if (only_pass >= 0 && pass != only_pass) continue;
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = sa[pass].c[k]; v[4 + k] = sb[pass].c[k]; }
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
}


// The fused Poseidon2 pieces (zc_poseidon2.hpp): one workgroup = 256 row pairs of one piece at one node, like the interpreter's
// workgroups and into the same partial-sum layout; descriptor flags bit 1 marks a macro piece, bits 8..11 its index q, `pad`
// its first main column. Every column of the permutation is loaded exactly once per piece; the loads a piece OWNS carry the
// GKR-opening batching term, so the interpreter's pieces never touch those columns for it (the planner pre-marks them).
constexpr uint32_t ZC_DESC_MACRO = 2u;
// KIND of a launch that carries the pieces of BOTH septic kinds (they are adjacent block ranges; the kind comes from the descriptor): in
// the small rounds every launch is at its latency floor and the two septic launches would share a hardware queue (a process has four)
constexpr uint32_t ZC_MACRO_BOTH_SEPTIC = 4u;
constexpr uint32_t ZC_MACRO_KINDS = 8;        // kinds 1..3, the launch shape 4, Keccak = 5, MulOperation products = 6, polynomial identities = 7
constexpr uint32_t ZC_POLY_WAVE_MAX_TERMS = 4096;    // row pairs of the tallest chip with polynomial identities below which a round runs them one wave per pair
constexpr uint32_t ZC_RANGE_CORNERS = ZC_MACRO_KINDS;   // (not a hint kind: the block range of zc_biv_corner_kernel in a bivariate plan)
template <bool FIRST, uint32_t KIND>
__global__ __launch_bounds__(256) void zc_macro_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                       uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base,
                                                       const p2::RoundConstants* __restrict__ rc_p) {
    using K = KT<FIRST>;
    using F = typename std::conditional<FIRST, P2Base, P2Ext>::type;
    using T = typename K::T;
    __shared__ uint32_t red[32];
    if (threadIdx.x < 32) red[threadIdx.x] = 0;
    const uint32_t bid = block_base + blockIdx.x / 3u;
    const int pass = (int)(blockIdx.x % 3u), t = 2 * pass;
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const uint32_t q = (d.flags >> 8) & 15u, base_col = d.pad;       // (every descriptor of this launch has kind KIND)
    const auto* rc = (const p2::RoundConstants __attribute__((address_space(4)))*)(uintptr_t)rc_p;    // wave-uniform: scalar loads
    __syncthreads();
    const uint32_t terms = (d.rows + 1) / 2;
    const bool gkr = !FIRST && pass < 2;
    kb::Ext sa = kb::ext_zero(), sb = kb::ext_zero();
    if (!(FIRST && pass == 0))                         // round 0, node 0: the constraints vanish and the GKR pass is the interpreter's
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < terms; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, terms); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        kb::Ext va = kb::ext_zero(), vb = kb::ext_zero();
        auto ld_at = [&](uint32_t col, bool owned) -> T {
            const T v = leaf<FIRST>(d.main, col, d.rows, i, t);
            if (gkr && owned) vb = kb::ext_add(vb, K::scale(load_ext_aos(d.gkr_pows, col), v));
            return v;
        };
        auto ld = [&](uint32_t c, bool owned) -> T { return ld_at(base_col + c, owned); };
        auto sink = [&](uint32_t j, const T& v) { va = kb::ext_add(va, K::scale(load_ext_aos(d.alpha_pows, d.alpha_off + j), v)); };
        auto alpha = [&](uint32_t j) -> kb::Ext { return load_ext_aos(d.alpha_pows, d.alpha_off + j); };
        auto emit = [&](const kb::Ext& v) { va = kb::ext_add(va, v); };
        if constexpr (KIND == ZC_HINT_POSEIDON2) zc_p2_piece<F>(q, rc, ld, sink);
        else if constexpr (KIND == ZC_HINT_KECCAK) zc_keccak_piece<F>(q, ld, sink);
        else if constexpr (KIND == ZC_HINT_MUL) zc_mul_piece<F>(q, ld, [&](uint32_t c, bool owned) -> T { return ld_at(d.aux0 + c, owned); }, sink);
        else if constexpr (KIND == ZC_HINT_SEPTIC_CURVE) zc_septic_curve_piece_w<F, K>(q, ld, alpha, emit);
        else if (KIND == ZC_MACRO_BOTH_SEPTIC && ((d.flags >> 12) & 15u) == ZC_HINT_SEPTIC_CURVE) zc_septic_curve_piece_w<F, K>(q, ld, alpha, emit);   // (wave-uniform)
        else zc_septic_sum_piece_w<F, K>(q, ld, [&](uint32_t c, bool owned) -> T { return ld_at(d.aux0 + c, owned); },
                                         [&]() -> T { return ld_at(d.aux1, false); }, alpha, emit);
        sa = kb::ext_add(sa, kb::ext_mul(va, e));
        sb = kb::ext_add(sb, kb::ext_mul(vb, e));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = sa.c[k]; v[4 + k] = sb.c[k]; }
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

// ---- bivariate kernels. Interpreter: blockIdx.x = 3 b + g -> (block b of `block_pairs` row QUADS of one chunk, node group g =
// nodes 4 g .. 4 g + 3, four node values per register: KT4). A node's slot is [A | B] like the single-round kernels': A = sum eq
// C(node e); B = the GKR batching term's corner sum (X, Y) = (e >> 1, e & 1) for e < 4 from the chip's first chunk (zero
// elsewhere). partial[(12 bid + e) * 8 ..).
constexpr uint32_t ZC_BIV_GROUPS = 3;
template <int MAXR, bool STAGED>
__global__ __launch_bounds__(256) void zc_biv_round_kernel(const ZcDesc* __restrict__ descs, int n_descs,
                                                           const uint32_t* __restrict__ eq, uint32_t eq_len,
                                                           const uint32_t* __restrict__ publics, uint32_t* __restrict__ partial,
                                                           uint32_t rf_off, uint32_t block_base) {
    using K = KT<true>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* red = lds;                                   // [4][8] reduction scratch
    uint4* lprog = reinterpret_cast<uint4*>(lds + 32);
    RegFile<false, MAXR> reg;                              // four base-field values per register (KT4)
    if constexpr (MAXR == 0) {
        reg.base = (decltype(reg.base))(lds + rf_off) + threadIdx.x;
        reg.stride = blockDim.x;
    }
    if (threadIdx.x < 32) red[threadIdx.x] = 0;
    const uint32_t bid = block_base + blockIdx.x / ZC_BIV_GROUPS;
    const uint32_t grp = blockIdx.x % ZC_BIV_GROUPS;
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    if constexpr (STAGED) {
        const uint4* src = reinterpret_cast<const uint4*>(d.prog);
        for (uint32_t k = threadIdx.x; k < d.n_instr; k += blockDim.x) lprog[k] = src[k];
    }
    __syncthreads();
    const uint32_t quads = (d.rows + 3) / 4;
    const bool corners = grp == 0 && (d.flags & 1u);
    kb::Ext sa[4], sb[4];
#pragma unroll
    for (int n = 0; n < 4; n++) { sa[n] = kb::ext_zero(); sb[n] = kb::ext_zero(); }
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < quads; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, quads); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        kb::Ext va[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
        if constexpr (STAGED) (void)run_program<true, MAXR, true>(reg, (const zc_word_t*)lprog, d, publics, i, (int)grp, false, va);
        else (void)run_program<true, MAXR, true>(reg, (zc_const_prog_t)(uintptr_t)d.prog, d, publics, i, (int)grp, false, va);
#pragma unroll
        for (int n = 0; n < 4; n++) sa[n] = kb::ext_add(sa[n], kb::ext_mul(va[n], e));
        if (corners) {                                      // row 4 i + n of every column, weighted by the GKR powers
            kb::Ext vb[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
            for (uint32_t c = 0; c < d.main_w + d.prep_w; c++) {
                const kb::Ext pw = load_ext_aos(d.gkr_pows, c);
                const uint32_t* tbl = c < d.main_w ? d.main : d.prep;
                const uint32_t col = c < d.main_w ? c : c - d.main_w;
#pragma unroll
                for (uint32_t n = 0; n < 4; n++)
                    if (4 * i + n < d.rows) vb[n] = kb::ext_add(vb[n], K::scale(pw, K::load(tbl, col, d.rows, 4 * i + n)));
            }
#pragma unroll
            for (int n = 0; n < 4; n++) sb[n] = kb::ext_add(sb[n], kb::ext_mul(vb[n], e));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = sa[n].c[k]; v[4 + k] = sb[n].c[k]; }
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
            partial[((size_t)bid * ZC_BIV_NODES + 4 * grp + n) * 8 + k] = kb::add(kb::add(red[k], red[8 + k]), kb::add(red[16 + k], red[24 + k]));
        }
    }
}

// The GKR batching term's corner sums of the bivariate rounds: B_n = sum_q eq(q) sum_c gkr_pow[c] * column c at row 4 q + n.
// Until round 6 the first chunk's node-group-0 workgroups of a chip walked ALL its columns for them, one after the other per lane:
// for the 2,640-column Keccak chip 400,000 dependent instructions on 477 waves — 6.2 ms, the longest launch of a Keccak shard's
// zerocheck by a factor of two, with the device idle around it. Here a workgroup takes 256 quads x ZC_CORNER_COLS columns:
// blockIdx.x = block of the launch's range; d.aux0 / d.aux1 = the slice [c0, c1) of the chip's main-then-preprocessed columns.
// Writes the B half of nodes 0..3 (what the interpreter's first chunk used to leave) and zeros everywhere else of its slots.
constexpr uint32_t ZC_CORNER_COLS = 32;
__global__ __launch_bounds__(256) void zc_biv_corner_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                            uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base) {
    using K = KT<true>;
    __shared__ uint32_t red[4 * 16];
    const uint32_t bid = block_base + blockIdx.x;
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const uint32_t c0 = d.aux0, c1 = d.aux1;
    const uint32_t quads = (d.rows + 3) / 4;
    kb::Ext sb[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < quads; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, quads); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        kb::Ext vb[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
#pragma unroll 4
        for (uint32_t c = c0; c < c1; c++) {
            const kb::Ext pw = load_ext_aos(d.gkr_pows, c);
            const uint32_t* tbl = c < d.main_w ? d.main : d.prep;
            const uint32_t col = c < d.main_w ? c : c - d.main_w;
#pragma unroll
            for (uint32_t n = 0; n < 4; n++)
                if (4 * i + n < d.rows) vb[n] = kb::ext_add(vb[n], K::scale(pw, K::load(tbl, col, d.rows, 4 * i + n)));
        }
#pragma unroll
        for (int n = 0; n < 4; n++) sb[n] = kb::ext_add(sb[n], kb::ext_mul(vb[n], e));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = zc_wave_sum(sb[n].c[k]);
            if (lane == 0) red[wave * 16 + 4 * n + k] = w;
        }
    __syncthreads();
    if (threadIdx.x < ZC_BIV_NODES * 8) {
        const uint32_t node = threadIdx.x / 8, k = threadIdx.x % 8;
        uint32_t v = 0;
        if (node < 4 && k >= 4) {
            const uint32_t j = 4 * node + (k - 4);
            v = kb::add(kb::add(red[j], red[16 + j]), kb::add(red[32 + j], red[48 + j]));
        }
        partial[((size_t)bid * ZC_BIV_NODES + node) * 8 + k] = v;
    }
}

// the fused pieces on the bivariate grid (base-field arithmetic: the pieces' P2Base forms; no GKR term here)
template <uint32_t KIND>
__global__ __launch_bounds__(256) void zc_biv_macro_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                           uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base,
                                                           const p2::RoundConstants* __restrict__ rc_p) {
    using K = KT<true>;
    __shared__ uint32_t red[32];
    if (threadIdx.x < 32) red[threadIdx.x] = 0;
    const uint32_t bid = block_base + blockIdx.x / (uint32_t)ZC_BIV_NODES;
    const uint32_t node = blockIdx.x % (uint32_t)ZC_BIV_NODES;
    const ZcBivNode nd = zc_biv_node(node);
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const uint32_t q = (d.flags >> 8) & 15u, base_col = d.pad;
    const auto* rc = (const p2::RoundConstants __attribute__((address_space(4)))*)(uintptr_t)rc_p;
    __syncthreads();
    const uint32_t quads = (d.rows + 3) / 4;
    kb::Ext sa = kb::ext_zero();
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < quads; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, quads); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        kb::Ext va = kb::ext_zero();
        auto ld_at = [&](uint32_t col, bool) -> uint32_t { return zc_biv_leaf(d.main, col, d.rows, i, nd); };
        auto ld = [&](uint32_t c, bool owned) -> uint32_t { return ld_at(base_col + c, owned); };
        auto sink = [&](uint32_t j, const uint32_t& v) { va = kb::ext_add(va, K::scale(load_ext_aos(d.alpha_pows, d.alpha_off + j), v)); };
        auto alpha = [&](uint32_t j) -> kb::Ext { return load_ext_aos(d.alpha_pows, d.alpha_off + j); };
        auto emit = [&](const kb::Ext& v) { va = kb::ext_add(va, v); };
        if constexpr (KIND == ZC_HINT_POSEIDON2) zc_p2_piece<P2Base>(q, rc, ld, sink);
        else if constexpr (KIND == ZC_HINT_KECCAK) zc_keccak_piece<P2Base>(q, ld, sink);
        else if constexpr (KIND == ZC_HINT_MUL) zc_mul_piece<P2Base>(q, ld, [&](uint32_t c, bool owned) -> uint32_t { return ld_at(d.aux0 + c, owned); }, sink);
        else if constexpr (KIND == ZC_HINT_SEPTIC_CURVE) zc_septic_curve_piece_w<P2Base, K>(q, ld, alpha, emit);
        else zc_septic_sum_piece_w<P2Base, K>(q, ld, [&](uint32_t c, bool owned) -> uint32_t { return ld_at(d.aux0 + c, owned); },
                                              [&]() -> uint32_t { return ld_at(d.aux1, false); }, alpha, emit);
        sa = kb::ext_add(sa, kb::ext_mul(va, e));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = zc_wave_sum(sa.c[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) red[wave * 8 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const uint32_t k = threadIdx.x;
        partial[((size_t)bid * ZC_BIV_NODES + node) * 8 + k] = k < 4 ? kb::add(kb::add(red[k], red[8 + k]), kb::add(red[16 + k], red[24 + k])) : 0u;
    }
}

// ---- polynomial identities (zc_poly.hpp, hint kind 7): sum_t A_t B_t + R with affine forms A_t, B_t, R whose coefficients the host
// collapsed for this proof's alpha (table at d.prog: header, then 8-word entries). An affine form's value at a node of a row pair is
// the interpolation of its values on the two rows, so ONE workgroup (blockIdx.x = block) loads every column of its 256 row pairs once
// and leaves the sums of all three nodes: partial slots as the per-node kernels write them. The columns the identity owns (carry and
// witness limbs: nothing else reads them) carry their GKR batching term in the extension rounds.
struct ZcPolyTable {
    zc_const_words_t tb;
    __device__ __forceinline__ uint32_t word(uint32_t off) const { return tb[off]; }
    __device__ __forceinline__ kb::Ext coef(uint32_t off) const { return kb::Ext{{tb[off + 4], tb[off + 5], tb[off + 6], tb[off + 7]}}; }
};
__device__ __forceinline__ kb::Ext zc_ext_times_pow2(kb::Ext v, uint32_t k) {          // k in {0, 1, 2, 4, 8, 16}: compile-time after unrolling
    if (k == 0) return kb::ext_zero();
    for (uint32_t m = 1; m < k; m <<= 1) v = kb::ext_add(v, v);
    return v;
}
template <bool FIRST>
__global__ __launch_bounds__(256) void zc_poly_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                      uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base) {
    using K = KT<FIRST>;
    using T = typename K::T;
    __shared__ uint32_t red[4 * 24];
    const uint32_t bid = block_base + blockIdx.x;
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const ZcPolyTable tab{(zc_const_words_t)(uintptr_t)d.prog};
    const uint32_t n_terms = tab.word(0), n_rest = tab.word(1), n_owned = tab.word(2);
    const uint32_t terms = (d.rows + 1) / 2;
    kb::Ext sa[3] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero()}, sb[2] = {kb::ext_zero(), kb::ext_zero()};
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < terms; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, terms); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        const bool has1 = 2 * i + 1 < d.rows;
        uint32_t off = ZC_POLY_HDR;
        // the values of one affine form on the two rows of the pair (its constant entry first)
        auto form = [&](uint32_t n, kb::Ext& f0, kb::Ext& f1) {
            f0 = f1 = tab.coef(off);
            off += ZC_POLY_ENTRY;
#pragma unroll 4
            for (uint32_t k = 0; k < n; k++, off += ZC_POLY_ENTRY) {
                const uint32_t col = tab.word(off);
                const kb::Ext c = tab.coef(off);
                const T x0 = K::load(d.main, col, d.rows, 2 * i);
                const T x1 = has1 ? K::load(d.main, col, d.rows, 2 * i + 1) : K::zero();
                f0 = kb::ext_add(f0, K::scale(c, x0));
                f1 = kb::ext_add(f1, K::scale(c, x1));
            }
        };
        kb::Ext v0 = kb::ext_zero(), v2 = kb::ext_zero(), v4 = kb::ext_zero();      // eq * C at the three nodes
        for (uint32_t t = 0; t < n_terms; t++) {
            kb::Ext a0, a1, b0, b1;
            form(tab.word(4 + 3 * t), a0, a1);
            form(tab.word(5 + 3 * t), b0, b1);
            a0 = kb::ext_mul(a0, e); a1 = kb::ext_mul(a1, e);
            const kb::Ext da = kb::ext_sub(a1, a0), db = kb::ext_sub(b1, b0);
            const kb::Ext da2 = kb::ext_add(da, da), db2 = kb::ext_add(db, db);
            const kb::Ext a2 = kb::ext_add(a0, da2), b2 = kb::ext_add(b0, db2);
            kb::Ext p0 = FIRST ? kb::ext_zero() : kb::ext_mul(a0, b0), p2 = kb::ext_mul(a2, b2), p4 = kb::ext_mul(kb::ext_add(a2, da2), kb::ext_add(b2, db2));
            const uint32_t n2 = tab.word(6 + 3 * t);
            if (n2 != ZC_POLY_NONE) {                                               // (wave-uniform)
                kb::Ext c0, c1;
                form(n2, c0, c1);
                const kb::Ext dc = kb::ext_sub(c1, c0), dc2 = kb::ext_add(dc, dc), c2 = kb::ext_add(c0, dc2);
                if (!FIRST) p0 = kb::ext_mul(p0, c0);
                p2 = kb::ext_mul(p2, c2);
                p4 = kb::ext_mul(p4, kb::ext_add(c2, dc2));
            }
            v0 = kb::ext_add(v0, p0); v2 = kb::ext_add(v2, p2); v4 = kb::ext_add(v4, p4);
        }
        kb::Ext r0, r1, g0 = kb::ext_zero(), g1 = kb::ext_zero();
        form(n_rest, r0, r1);
#pragma unroll 2
        for (uint32_t k = 0; k < n_owned; k++, off += ZC_POLY_ENTRY) {
            const uint32_t col = tab.word(off);
            const kb::Ext c = tab.coef(off);
            const T x0 = K::load(d.main, col, d.rows, 2 * i);
            const T x1 = has1 ? K::load(d.main, col, d.rows, 2 * i + 1) : K::zero();
            r0 = kb::ext_add(r0, K::scale(c, x0));
            r1 = kb::ext_add(r1, K::scale(c, x1));
            if (!FIRST) {
                const kb::Ext gp = load_ext_aos(d.gkr_pows, col);
                g0 = kb::ext_add(g0, K::scale(gp, x0));
                g1 = kb::ext_add(g1, K::scale(gp, x1));
            }
        }
        r0 = kb::ext_mul(r0, e); r1 = kb::ext_mul(r1, e);
        const kb::Ext dr = kb::ext_sub(r1, r0), dr2 = kb::ext_add(dr, dr), r2 = kb::ext_add(r0, dr2);
        if (!FIRST) sa[0] = kb::ext_add(sa[0], kb::ext_add(v0, r0));
        sa[1] = kb::ext_add(sa[1], kb::ext_add(v2, r2));
        sa[2] = kb::ext_add(sa[2], kb::ext_add(v4, kb::ext_add(r2, dr2)));
        if (!FIRST) {
            g0 = kb::ext_mul(g0, e); g1 = kb::ext_mul(g1, e);
            const kb::Ext dg = kb::ext_sub(g1, g0);
            sb[0] = kb::ext_add(sb[0], g0);
            sb[1] = kb::ext_add(sb[1], kb::ext_add(g0, kb::ext_add(dg, dg)));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v[24];
#pragma unroll
    for (int pass = 0; pass < 3; pass++)
#pragma unroll
        for (int k = 0; k < 4; k++) { v[pass * 8 + k] = zc_wave_sum(sa[pass].c[k]); v[pass * 8 + 4 + k] = pass < 2 ? zc_wave_sum(sb[pass].c[k]) : 0u; }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 24; k++) red[wave * 24 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        const uint32_t k = threadIdx.x;
        uint32_t acc = red[k];
        for (uint32_t w = 1; w < blockDim.x / 64; w++) acc = kb::add(acc, red[w * 24 + k]);
        partial[((size_t)bid * 3 + k / 8) * 8 + (k & 7u)] = acc;
    }
}

// The same in the SMALL extension rounds, one WAVE per row pair: a lane of zc_poly_kernel walks every entry of every form of its
// pair — ~350 dependent load-multiply steps for a 48-limb field operation —, and once a round has only a few thousand pairs that
// walk is the round: 400 us per round whatever its size, 6.5 ms of a bls12-381 Fp shard's 23.8 ms of zerocheck. Here the 64 lanes
// of a wave take the entries of a form 64 at a time and the form is a wave sum; workgroup = 4 pairs (d.block_pairs = 4).
__global__ __launch_bounds__(256) void zc_poly_wave_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                           uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base) {
    using K = KT<false>;
    __shared__ uint32_t red[4 * 24];
    const uint32_t bid = block_base + blockIdx.x;
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const ZcPolyTable tab{(zc_const_words_t)(uintptr_t)d.prog};
    const zc_global_words_t tg = (zc_global_words_t)d.prog;           // (entries are read per lane)
    const zc_global_words_t gp = (zc_global_words_t)d.gkr_pows;
    const uint32_t n_terms = tab.word(0), n_rest = tab.word(1), n_owned = tab.word(2);
    const uint32_t terms = (d.rows + 1) / 2;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    kb::Ext sa[3] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero()}, sb[2] = {kb::ext_zero(), kb::ext_zero()};
    auto wsum = [&](kb::Ext& v) {
#pragma unroll
        for (int k = 0; k < 4; k++) v.c[k] = zc_wave_sum(v.c[k]);
    };
    for (uint32_t i = (bid - d.block_start) * 4 + wave; i < terms; i += d.n_blocks * 4) {     // (wave-uniform)
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        const bool has1 = 2 * i + 1 < d.rows;
        uint32_t off = ZC_POLY_HDR;
        // one affine form on the two rows: `n` entries behind an optional constant entry; with_gkr: the owned columns' batching term too
        auto form = [&](uint32_t n, bool with_const, kb::Ext& f0, kb::Ext& f1, kb::Ext* g0, kb::Ext* g1) {
            const uint32_t first = off + (with_const ? ZC_POLY_ENTRY : 0u);
            kb::Ext p0 = kb::ext_zero(), p1 = kb::ext_zero(), q0 = kb::ext_zero(), q1 = kb::ext_zero();
            for (uint32_t k = lane; k < n; k += 64) {
                const uint32_t o = first + ZC_POLY_ENTRY * k;
                const uint32_t col = tg[o];
                const kb::Ext c{{tg[o + 4], tg[o + 5], tg[o + 6], tg[o + 7]}};
                const kb::Ext x0 = K::load(d.main, col, d.rows, 2 * i);
                const kb::Ext x1 = has1 ? K::load(d.main, col, d.rows, 2 * i + 1) : kb::ext_zero();
                p0 = kb::ext_add(p0, kb::ext_mul(x0, c));
                p1 = kb::ext_add(p1, kb::ext_mul(x1, c));
                if (g0) {
                    const kb::Ext w{{gp[4 * col], gp[4 * col + 1], gp[4 * col + 2], gp[4 * col + 3]}};
                    q0 = kb::ext_add(q0, kb::ext_mul(x0, w));
                    q1 = kb::ext_add(q1, kb::ext_mul(x1, w));
                }
            }
            wsum(p0); wsum(p1);
            if (with_const) { const kb::Ext c0 = tab.coef(off); p0 = kb::ext_add(p0, c0); p1 = kb::ext_add(p1, c0); }
            f0 = p0; f1 = p1;
            if (g0) { wsum(q0); wsum(q1); *g0 = q0; *g1 = q1; }
            off = first + ZC_POLY_ENTRY * n;
        };
        kb::Ext v0 = kb::ext_zero(), v2 = kb::ext_zero(), v4 = kb::ext_zero();
        for (uint32_t t = 0; t < n_terms; t++) {
            kb::Ext a0, a1, b0, b1;
            form(tab.word(4 + 3 * t), true, a0, a1, nullptr, nullptr);
            form(tab.word(5 + 3 * t), true, b0, b1, nullptr, nullptr);
            a0 = kb::ext_mul(a0, e); a1 = kb::ext_mul(a1, e);
            const kb::Ext da = kb::ext_sub(a1, a0), db = kb::ext_sub(b1, b0);
            const kb::Ext da2 = kb::ext_add(da, da), db2 = kb::ext_add(db, db);
            const kb::Ext a2 = kb::ext_add(a0, da2), b2 = kb::ext_add(b0, db2);
            kb::Ext p0 = kb::ext_mul(a0, b0), p2 = kb::ext_mul(a2, b2), p4 = kb::ext_mul(kb::ext_add(a2, da2), kb::ext_add(b2, db2));
            const uint32_t n2 = tab.word(6 + 3 * t);
            if (n2 != ZC_POLY_NONE) {
                kb::Ext c0, c1;
                form(n2, true, c0, c1, nullptr, nullptr);
                const kb::Ext dc = kb::ext_sub(c1, c0), dc2 = kb::ext_add(dc, dc), c2 = kb::ext_add(c0, dc2);
                p0 = kb::ext_mul(p0, c0); p2 = kb::ext_mul(p2, c2); p4 = kb::ext_mul(p4, kb::ext_add(c2, dc2));
            }
            v0 = kb::ext_add(v0, p0); v2 = kb::ext_add(v2, p2); v4 = kb::ext_add(v4, p4);
        }
        kb::Ext r0, r1, o0, o1, g0, g1;
        form(n_rest, true, r0, r1, nullptr, nullptr);
        form(n_owned, false, o0, o1, &g0, &g1);
        r0 = kb::ext_mul(kb::ext_add(r0, o0), e); r1 = kb::ext_mul(kb::ext_add(r1, o1), e);
        const kb::Ext dr = kb::ext_sub(r1, r0), dr2 = kb::ext_add(dr, dr), r2 = kb::ext_add(r0, dr2);
        sa[0] = kb::ext_add(sa[0], kb::ext_add(v0, r0));
        sa[1] = kb::ext_add(sa[1], kb::ext_add(v2, r2));
        sa[2] = kb::ext_add(sa[2], kb::ext_add(v4, kb::ext_add(r2, dr2)));
        g0 = kb::ext_mul(g0, e); g1 = kb::ext_mul(g1, e);
        const kb::Ext dg = kb::ext_sub(g1, g0);
        sb[0] = kb::ext_add(sb[0], g0);
        sb[1] = kb::ext_add(sb[1], kb::ext_add(g0, kb::ext_add(dg, dg)));
    }
    if (lane == 0) {                                                   // (every lane of a wave holds the same sums)
#pragma unroll
        for (int pass = 0; pass < 3; pass++)
#pragma unroll
            for (int k = 0; k < 4; k++) { red[wave * 24 + pass * 8 + k] = sa[pass].c[k]; red[wave * 24 + pass * 8 + 4 + k] = pass < 2 ? sb[pass].c[k] : 0u; }
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        const uint32_t k = threadIdx.x;
        const uint32_t acc = kb::add(kb::add(red[k], red[24 + k]), kb::add(red[48 + k], red[72 + k]));
        partial[((size_t)bid * 3 + k / 8) * 8 + (k & 7u)] = acc;
    }
}

// The bivariate rounds: row quads, the twelve nodes of the grid from the forms' values on the four rows (base-field words).
__global__ __launch_bounds__(256) void zc_biv_poly_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                          uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base) {
    __shared__ uint32_t red[4 * 48];
    const uint32_t bid = block_base + blockIdx.x;
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const ZcPolyTable tab{(zc_const_words_t)(uintptr_t)d.prog};
    const uint32_t n_terms = tab.word(0), n_rest = tab.word(1), n_owned = tab.word(2);
    const uint32_t quads = (d.rows + 3) / 4;
    kb::Ext sa[ZC_BIV_NODES];
#pragma unroll
    for (int n = 0; n < ZC_BIV_NODES; n++) sa[n] = kb::ext_zero();
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < quads; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, quads); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        const uint32_t r = 4 * i;
        uint32_t off = ZC_POLY_HDR;
        auto form = [&](uint32_t n, kb::Ext (&f)[4]) {              // f: the form on rows r .. r + 3 = (X, Y) = (0,0) (0,1) (1,0) (1,1)
            f[0] = f[1] = f[2] = f[3] = tab.coef(off);
            off += ZC_POLY_ENTRY;
#pragma unroll 2
            for (uint32_t k = 0; k < n; k++, off += ZC_POLY_ENTRY) {
                const uint32_t col = tab.word(off);
                const kb::Ext c = tab.coef(off);
                const zc_global_words_t g = (zc_global_words_t)d.main + (size_t)col * d.rows;
                const uint32_t x00 = g[r], x01 = r + 1 < d.rows ? g[r + 1] : 0u, x10 = r + 2 < d.rows ? g[r + 2] : 0u, x11 = r + 3 < d.rows ? g[r + 3] : 0u;
                f[0] = kb::ext_add(f[0], kb::ext_mul_base(c, x00));
                f[1] = kb::ext_add(f[1], kb::ext_mul_base(c, x01));
                f[2] = kb::ext_add(f[2], kb::ext_mul_base(c, x10));
                f[3] = kb::ext_add(f[3], kb::ext_mul_base(c, x11));
            }
        };
        // f -> (f00, dX, dY, dXY): the form at node (X, Y) is f00 + X dX + Y dY + X Y dXY
        auto slopes = [&](kb::Ext (&f)[4]) {
            const kb::Ext dy = kb::ext_sub(f[1], f[0]), dx = kb::ext_sub(f[2], f[0]);
            f[3] = kb::ext_sub(kb::ext_sub(f[3], f[2]), dy);
            f[1] = dx; f[2] = dy;
        };
        auto at = [&](const kb::Ext (&f)[4], const ZcBivNode& nd) -> kb::Ext {
            return kb::ext_add(kb::ext_add(f[0], zc_ext_times_pow2(f[1], nd.cx)), kb::ext_add(zc_ext_times_pow2(f[2], nd.cy), zc_ext_times_pow2(f[3], nd.cxy)));
        };
        for (uint32_t t = 0; t < n_terms; t++) {
            kb::Ext a[4], b[4];
            form(tab.word(4 + 3 * t), a);
            form(tab.word(5 + 3 * t), b);
#pragma unroll
            for (int k = 0; k < 4; k++) a[k] = kb::ext_mul(a[k], e);
            slopes(a); slopes(b);
            const uint32_t n2 = tab.word(6 + 3 * t);
            if (n2 == ZC_POLY_NONE) {                                               // (wave-uniform)
#pragma unroll
                for (int n = 0; n < ZC_BIV_NODES; n++) {
                    const ZcBivNode nd = zc_biv_node(n);
                    sa[n] = kb::ext_add(sa[n], kb::ext_mul(at(a, nd), at(b, nd)));
                }
            } else {
                kb::Ext c[4];
                form(n2, c);
                slopes(c);
#pragma unroll
                for (int n = 0; n < ZC_BIV_NODES; n++) {
                    const ZcBivNode nd = zc_biv_node(n);
                    sa[n] = kb::ext_add(sa[n], kb::ext_mul(kb::ext_mul(at(a, nd), at(b, nd)), at(c, nd)));
                }
            }
        }
        kb::Ext rr[4];
        form(n_rest + n_owned, rr);                                 // (the owned columns follow the rest's: no GKR term in these rounds)
#pragma unroll
        for (int k = 0; k < 4; k++) rr[k] = kb::ext_mul(rr[k], e);
        slopes(rr);
#pragma unroll
        for (int n = 0; n < ZC_BIV_NODES; n++) sa[n] = kb::ext_add(sa[n], at(rr, zc_biv_node(n)));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v[48];
#pragma unroll
    for (int n = 0; n < ZC_BIV_NODES; n++)
#pragma unroll
        for (int k = 0; k < 4; k++) v[n * 4 + k] = zc_wave_sum(sa[n].c[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 48; k++) red[wave * 48 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 48) {
        const uint32_t k = threadIdx.x;
        uint32_t acc = red[k];
        for (uint32_t w = 1; w < blockDim.x / 64; w++) acc = kb::add(acc, red[w * 48 + k]);
        partial[((size_t)bid * ZC_BIV_NODES + k / 4) * 8 + (k & 3u)] = acc;
        partial[((size_t)bid * ZC_BIV_NODES + k / 4) * 8 + 4 + (k & 3u)] = 0u;
    }
}

// The Keccak pieces in the extension rounds, the THREE nodes of a row pair per pass (SP1HIP_ZC_KECCAK3=1; off by default, see below): the pieces are
// bound by the bandwidth of their column loads, and one node per workgroup reads both rows of every column three times. Here a
// lane loads the two rows once and carries the values at t = 0, 2, 4 through the piece (element-wise arithmetic on three extension
// values: 12 VGPRs per live value). blockIdx.x = block; partial slots of the three nodes as the per-node kernels write them.
// Measured (round 5, precompile shard): 250 VGPRs, two waves per SIMD, zerocheck rounds 31.0 ms against 22.0 ms for one node per
// workgroup -- the saved loads do not pay for the lost occupancy, so the per-node kernels stay the default.
struct E3 { kb::Ext n[3]; };
struct P2Ext3 {
    using T = E3;
    static __device__ __forceinline__ T add(const T& a, const T& b) { return T{{kb::ext_add(a.n[0], b.n[0]), kb::ext_add(a.n[1], b.n[1]), kb::ext_add(a.n[2], b.n[2])}}; }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return T{{kb::ext_sub(a.n[0], b.n[0]), kb::ext_sub(a.n[1], b.n[1]), kb::ext_sub(a.n[2], b.n[2])}}; }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return T{{kb::ext_mul(a.n[0], b.n[0]), kb::ext_mul(a.n[1], b.n[1]), kb::ext_mul(a.n[2], b.n[2])}}; }
    static __device__ __forceinline__ T addc(T a, uint32_t c) {
#pragma unroll
        for (int k = 0; k < 3; k++) a.n[k].c[0] = kb::add(a.n[k].c[0], c);
        return a;
    }
    static __device__ __forceinline__ T mulc(const T& a, uint32_t c) { return T{{kb::ext_mul_base(a.n[0], c), kb::ext_mul_base(a.n[1], c), kb::ext_mul_base(a.n[2], c)}}; }
};
__global__ __launch_bounds__(256) void zc_keccak3_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                         uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base) {
    using K = KT<false>;
    __shared__ uint32_t red[4][24];
    const uint32_t bid = block_base + blockIdx.x;
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const uint32_t q = (d.flags >> 8) & 15u, base_col = d.pad;
    const uint32_t terms = (d.rows + 1) / 2;
    kb::Ext sa[3] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero()}, sb[2] = {kb::ext_zero(), kb::ext_zero()};
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < terms; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, terms); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        kb::Ext va[3] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero()}, vb[2] = {kb::ext_zero(), kb::ext_zero()};
        const bool odd = 2 * i + 1 < d.rows;
        auto ld = [&](uint32_t c, bool owned) -> E3 {
            const uint32_t col = base_col + c;
            const kb::Ext r0 = K::load(d.main, col, d.rows, 2 * i);
            const kb::Ext r1 = odd ? K::load(d.main, col, d.rows, 2 * i + 1) : kb::ext_zero();
            const kb::Ext slope = kb::ext_sub(r1, r0), s2 = kb::ext_add(slope, slope);
            E3 v;
            v.n[0] = r0;
            v.n[1] = kb::ext_add(s2, r0);
            v.n[2] = kb::ext_add(kb::ext_add(s2, s2), r0);
            if (owned) {                                   // the GKR-opening batching term at nodes 0 and 2 (g(4) = 2 g(2) - g(0))
                const kb::Ext pw = load_ext_aos(d.gkr_pows, col);
                vb[0] = kb::ext_add(vb[0], kb::ext_mul(v.n[0], pw));
                vb[1] = kb::ext_add(vb[1], kb::ext_mul(v.n[1], pw));
            }
            return v;
        };
        auto sink = [&](uint32_t j, const E3& v) {
            const kb::Ext a = load_ext_aos(d.alpha_pows, d.alpha_off + j);
#pragma unroll
            for (int n = 0; n < 3; n++) va[n] = kb::ext_add(va[n], kb::ext_mul(v.n[n], a));
        };
        zc_keccak_piece<P2Ext3>(q, ld, sink);
#pragma unroll
        for (int n = 0; n < 3; n++) sa[n] = kb::ext_add(sa[n], kb::ext_mul(va[n], e));
#pragma unroll
        for (int n = 0; n < 2; n++) sb[n] = kb::ext_add(sb[n], kb::ext_mul(vb[n], e));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // words: pass p -> [A_p (4) | B_p (4)], B_2 = 0
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t w = k < 4 ? sa[p].c[k] : (p < 2 ? sb[p].c[k - 4] : 0u);
            const uint32_t v = zc_wave_sum(w);
            if (lane == 0) red[wave][p * 8 + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < 24) {
        const uint32_t w = threadIdx.x;
        partial[(size_t)bid * 24 + w] = kb::add(kb::add(red[0][w], red[1][w]), kb::add(red[2][w], red[3][w]));
    }
}

// The Keccak pieces on the bivariate grid, FOUR nodes per pass (node group g = nodes 4 g .. 4 g + 3, like the interpreter's KT4
// passes): the pieces are bound by the bandwidth their column loads draw from the caches, and one node per workgroup reads the
// four rows of every column twelve times. Here the rows are loaded once per group and interpolated to the group's four nodes; the
// arithmetic is element-wise on the 4-vector. blockIdx.x = 3 b + g.
struct P2Base4 {
    using T = kb::Ext;                                   // four base-field node values
    static __device__ __forceinline__ T add(const T& a, const T& b) { return kb::ext_add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return kb::ext_sub(a, b); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return KT4::mul(a, b); }
    static __device__ __forceinline__ T addc(const T& a, uint32_t c) { return KC4::addc(a, c); }
    static __device__ __forceinline__ T mulc(const T& a, uint32_t c) { return kb::ext_mul_base(a, c); }
};
__global__ __launch_bounds__(256) void zc_biv_keccak_kernel(const ZcDesc* __restrict__ descs, int n_descs, const uint32_t* __restrict__ eq,
                                                            uint32_t eq_len, uint32_t* __restrict__ partial, uint32_t block_base) {
    using K = KT<true>;
    __shared__ uint32_t red[4][32];
    const uint32_t bid = block_base + blockIdx.x / ZC_BIV_GROUPS;
    const uint32_t grp = blockIdx.x % ZC_BIV_GROUPS;
    const ZcBivNode n0 = zc_biv_node(4 * grp), n1 = zc_biv_node(4 * grp + 1), n2 = zc_biv_node(4 * grp + 2), n3 = zc_biv_node(4 * grp + 3);
    const ZcDesc d = zc_find_desc(descs, n_descs, bid);
    const uint32_t q = (d.flags >> 8) & 15u, base_col = d.pad;
    const uint32_t quads = (d.rows + 3) / 4;
    kb::Ext sa[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
    for (uint32_t base = (bid - d.block_start) * d.block_pairs; base < quads; base += d.n_blocks * d.block_pairs)
    for (uint32_t i = base + threadIdx.x; i < min(base + d.block_pairs, quads); i += blockDim.x) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * eq_len + i];
        kb::Ext va[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
        auto ld = [&](uint32_t c, bool) -> kb::Ext {
            const zc_global_words_t g = (zc_global_words_t)d.main + (size_t)(base_col + c) * d.rows;
            const uint32_t r = 4 * i;
            const uint32_t r00 = g[r], r01 = r + 1 < d.rows ? g[r + 1] : 0u, r10 = r + 2 < d.rows ? g[r + 2] : 0u, r11 = r + 3 < d.rows ? g[r + 3] : 0u;
            return kb::Ext{{zc_biv_interp(r00, r01, r10, r11, n0), zc_biv_interp(r00, r01, r10, r11, n1), zc_biv_interp(r00, r01, r10, r11, n2),
                            zc_biv_interp(r00, r01, r10, r11, n3)}};
        };
        auto sink = [&](uint32_t j, const kb::Ext& v) {
            const kb::Ext a = load_ext_aos(d.alpha_pows, d.alpha_off + j);
#pragma unroll
            for (int n = 0; n < 4; n++) va[n] = kb::ext_add(va[n], K::scale(a, v.c[n]));
        };
        zc_keccak_piece<P2Base4>(q, ld, sink);
#pragma unroll
        for (int n = 0; n < 4; n++) sa[n] = kb::ext_add(sa[n], kb::ext_mul(va[n], e));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t v = zc_wave_sum(sa[n].c[k]);
            if (lane == 0) red[wave][n * 4 + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < 32) {
        const uint32_t n = threadIdx.x >> 3, k = threadIdx.x & 7u;
        const uint32_t w = n * 4 + (k & 3u);
        partial[((size_t)bid * ZC_BIV_NODES + 4 * grp + n) * 8 + k] =
            k < 4 ? kb::add(kb::add(red[0][w], red[1][w]), kb::add(red[2][w], red[3][w])) : 0u;
    }
}

// One workgroup per (range, node): the node's [A | B] summed over the range's blocks. Per range the output is A_0..11 (12 ext),
// the four corner sums B_0..3 (4 ext), eq[th] (1 ext): out[range][68] and, with a host slot, payload words [1 + 68 range ..);
// the last workgroup of the launch publishes `seq`. (One workgroup per range took 235 us for a chip of 12k blocks.)
constexpr uint32_t ZC_BIV_SUM_WORDS = 68;
__global__ __launch_bounds__(256) void zc_biv_reduce_kernel(const ZcChipRange* __restrict__ ranges, const uint32_t* __restrict__ partial,
                                                            const uint32_t* __restrict__ eq, uint32_t eq_len,
                                                            uint32_t* __restrict__ out, RoundSync rs, uint32_t seq) {
    __shared__ uint32_t acc[32][8];
    const uint32_t range = blockIdx.x / (uint32_t)ZC_BIV_NODES, node = blockIdx.x % (uint32_t)ZC_BIV_NODES;
    const ZcChipRange d = ranges[range];
    const uint32_t word = threadIdx.x & 7u, grp = threadIdx.x >> 3;       // 32 groups of 8 words
    {
        const uint32_t* p = partial + ((size_t)d.block_start * ZC_BIV_NODES + node) * 8 + word;
        uint32_t a[4] = {0, 0, 0, 0};
        uint32_t b = grp;
        for (; b + 96 < d.n_blocks; b += 128)
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = kb::add(a[u], p[(size_t)(b + 32 * u) * (ZC_BIV_NODES * 8)]);
        for (; b < d.n_blocks; b += 32) a[0] = kb::add(a[0], p[(size_t)b * (ZC_BIV_NODES * 8)]);
        acc[grp][word] = kb::add(kb::add(a[0], a[1]), kb::add(a[2], a[3]));
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        // words 0..3: A_node -> [4 node ..), words 4..7: B_node -> [48 + 4 node ..) for node < 4, words 8..11: eq[th] -> [64 ..) from node 0
        const uint32_t w = threadIdx.x;
        uint32_t val = 0, dst = 0xffffffffu;
        if (w < 8) {
            for (uint32_t g = 0; g < 32; g++) val = kb::add(val, acc[g][w]);
            if (w < 4) dst = 4 * node + w;
            else if (node < 4) dst = 48 + 4 * node + (w - 4);
        } else if (node == 0) {
            const uint32_t k = w - 8;
            val = d.th < eq_len ? eq[(size_t)k * eq_len + d.th] : 0u;
            dst = 64 + k;
        }
        if (dst != 0xffffffffu) {
            out[(size_t)range * ZC_BIV_SUM_WORDS + dst] = val;
            if (rs.host_slot != nullptr)
                __hip_atomic_store(const_cast<uint32_t*>(rs.host_slot) + 1 + (size_t)range * ZC_BIV_SUM_WORDS + dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (rs.host_slot == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && rs_ticket_is_last_acq_rel(rs.counter, blockIdx.x, gridDim.x)) rs_publish_seq(rs.host_slot, seq);
}

// One workgroup per chip: sums its workgroups' partials and forms (y0, y2, y4, eq[th]) -> out[chip][16].
template <bool FIRST>
__global__ __launch_bounds__(256) void zc_reduce_kernel(const ZcChipRange* __restrict__ ranges, const uint32_t* __restrict__ partial,
                                                        const uint32_t* __restrict__ eq, uint32_t eq_len,
                                                        uint32_t* __restrict__ out, RoundSync rs, uint32_t seq) {
    __shared__ uint32_t acc[10][24];
    const ZcChipRange d = ranges[blockIdx.x];
    uint32_t y0, y2, y4, e;
    zc_reduce_range<FIRST>(d, partial, eq, eq_len, acc, y0, y2, y4, e);
    if (threadIdx.x < 4) {
        const uint32_t k = threadIdx.x;
        uint32_t* o = out + (size_t)blockIdx.x * 16;
        o[k] = y0; o[4 + k] = y2; o[8 + k] = y4;
        o[12 + k] = e;
    }
    // the round's result goes to the host from HERE (payload words [1 + 16 chip ..)): system-scope stores, and below the
    // workgroup that arrives last publishes the sequence number — no mailbox kernel behind this one
    if (rs.host_slot != nullptr) zc_store_host_sums(rs.host_slot, blockIdx.x, y0, y2, y4, e);
    if (rs.host_slot == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && rs_ticket_is_last_acq_rel(rs.counter, blockIdx.x, gridDim.x)) rs_publish_seq(rs.host_slot, seq);
}

// out[i][c] = x + alpha (y - x), x = row 2i, y = row 2i + 1 (zero beyond the real rows); out is an ext table.
// One launch per round for every table of every chip. A workgroup owns ZC_FIX_ROWS consecutive rows of ONE column (8 per
// lane): the column comes from the block index with a multiply-high and the table from a binary search. The former form —
// one element per thread over the flattened table, 64-bit division and modulo per element, a linear scan of the ~45
// descriptors per workgroup — launched 786k workgroups for the first round's 2e8 elements and ran at 3.2 TB/s.
constexpr uint32_t ZC_FIX_ROWS = 2048;          // output rows of one column per workgroup (8 per lane)
template <bool FIRST>
__global__ __launch_bounds__(256) void zc_fix_kernel(const ZcFixDesc* __restrict__ descs, int n_descs, kb::Ext alpha) {
    using K = KT<FIRST>;
    int lo = 0, hi = n_descs - 1;                                    // last descriptor with block_start <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (__builtin_amdgcn_readfirstlane(descs[mid].block_start) <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ZcFixDesc d = descs[lo];
    const uint32_t out_rows = (d.rows + 1) / 2;
    const uint32_t lb = blockIdx.x - d.block_start;
    uint32_t c = d.bpc == 1 ? lb : __umulhi(lb, d.bpc_magic);       // floor(lb / bpc), at most 2 short
    uint32_t tile = lb - c * d.bpc;
    if (tile >= d.bpc) { tile -= d.bpc; c++; }
    if (tile >= d.bpc) { tile -= d.bpc; c++; }
    const uint32_t i0 = tile * ZC_FIX_ROWS, i1 = min(out_rows, i0 + ZC_FIX_ROWS);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256u) {
        typename K::T x = K::load(d.in, c, d.rows, 2 * i);
        typename K::T y = (2 * i + 1 < d.rows) ? K::load(d.in, c, d.rows, 2 * i + 1) : K::zero();
        const kb::Ext r = kb::ext_add(K::scale(alpha, K::sub(y, x)), K::to_ext(x));
#pragma unroll
        for (int q = 0; q < 4; q++) gptr(d.out)[((size_t)c * 4 + q) * out_rows + i] = r.c[q];
    }
}

// The table update behind the bivariate rounds: out[q][c] = T_q(X = a1, Y = a0) — the fold by the first challenge and then by
// the second, from the base-field rows (fix_last_variable.rs applied twice): one pass, 4 A bytes read and 4 A written instead of
// 4 A + 8 A read and 8 A + 4 A written by two updates. Descriptors as for zc_fix_kernel with out_rows = ceil(rows / 4).
__global__ __launch_bounds__(256) void zc_fix2_kernel(const ZcFixDesc* __restrict__ descs, int n_descs, kb::Ext a0, kb::Ext a1) {
    int lo = 0, hi = n_descs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (__builtin_amdgcn_readfirstlane(descs[mid].block_start) <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ZcFixDesc d = descs[lo];
    const uint32_t out_rows = (d.rows + 3) / 4;
    const uint32_t lb = blockIdx.x - d.block_start;
    uint32_t c = d.bpc == 1 ? lb : __umulhi(lb, d.bpc_magic);
    uint32_t tile = lb - c * d.bpc;
    if (tile >= d.bpc) { tile -= d.bpc; c++; }
    if (tile >= d.bpc) { tile -= d.bpc; c++; }
    const uint32_t i0 = tile * ZC_FIX_ROWS, i1 = min(out_rows, i0 + ZC_FIX_ROWS);
    const zc_global_words_t g = (zc_global_words_t)d.in + (size_t)c * d.rows;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256u) {
        const uint32_t r = 4 * i;
        const uint32_t r00 = g[r], r01 = r + 1 < d.rows ? g[r + 1] : 0u, r10 = r + 2 < d.rows ? g[r + 2] : 0u, r11 = r + 3 < d.rows ? g[r + 3] : 0u;
        const uint32_t dy = kb::sub(r01, r00), dx = kb::sub(r10, r00), dxy = kb::sub(kb::sub(r11, r10), dy);
        kb::Ext lo_row = kb::ext_mul_base(a0, dy);             // row 2 i of the once-folded table: r00 + a0 (r01 - r00)
        lo_row.c[0] = kb::add(lo_row.c[0], r00);
        kb::Ext slope = kb::ext_mul_base(a0, dxy);             // (row 2 i + 1) - (row 2 i) = (r10 - r00) + a0 ((r11 - r10) - (r01 - r00))
        slope.c[0] = kb::add(slope.c[0], dx);
        const kb::Ext res = kb::ext_add(lo_row, kb::ext_mul(slope, a1));
#pragma unroll
        for (int k = 0; k < 4; k++) gptr(d.out)[((size_t)c * 4 + k) * out_rows + i] = res.c[k];
    }
}

struct ZcGatherDesc { const uint32_t* src; uint32_t n_words, dst_off; };
__global__ __launch_bounds__(256) void zc_gather_kernel(const ZcGatherDesc* __restrict__ descs, uint32_t* __restrict__ out) {
    const ZcGatherDesc d = descs[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < d.n_words; i += 256) out[d.dst_off + i] = gptr(d.src)[i];
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

struct ZcMacro {                 // a hinted sub-AIR: its constraints are [first_constraint, first_constraint + n_constraints())
    uint32_t kind, base_col, first_constraint, aux0 = 0, aux1 = 0;
    uint32_t n_c = 0;            // kind 7 (a polynomial identity, zc_poly.hpp: ZcPlan::polys[aux0]): its number of constraints
    uint32_t n_constraints() const { return kind == ZC_HINT_POLY ? n_c : kind == ZC_HINT_POSEIDON2 ? ZC_P2_CONSTRAINTS : kind == ZC_HINT_KECCAK ? ZC_KK_CONSTRAINTS : kind == ZC_HINT_MUL ? ZC_MUL_CONSTRAINTS : kind == ZC_HINT_SEPTIC_CURVE ? 7u : 14u; }
    // pieces the kernels run (the septic kinds: weighted forms, zc_septic_*_piece_w) / pieces of the host model (per-coefficient forms)
    uint32_t n_pieces() const { return kind == ZC_HINT_POLY ? 1u : kind == ZC_HINT_POSEIDON2 ? ZC_P2_PIECES : kind == ZC_HINT_KECCAK ? ZC_KK_PIECES : kind == ZC_HINT_MUL ? ZC_MUL_PIECES : kind == ZC_HINT_SEPTIC_CURVE ? 2u : 4u; }
    uint32_t n_host_pieces() const { return kind == ZC_HINT_SEPTIC_CURVE ? 1u : kind == ZC_HINT_SEPTIC_SUM ? 2u : n_pieces(); }
    // the columns whose GKR-opening term the fused pieces carry: [lo, lo + n) (a polynomial identity: the list ZcPoly::owned instead)
    void owned(uint32_t* lo, uint32_t* n) const {
        if (kind == ZC_HINT_POLY) { *lo = 0; *n = 0; }
        else if (kind == ZC_HINT_POSEIDON2) { *lo = base_col; *n = ZC_P2_COLUMNS; }
        else if (kind == ZC_HINT_KECCAK) { *lo = base_col; *n = ZC_KK_COLUMNS; }
        else if (kind == ZC_HINT_MUL) { *lo = base_col + MUL_CARRY; *n = ZC_MUL_OWNED; }
        else if (kind == ZC_HINT_SEPTIC_CURVE) { *lo = base_col; *n = 14; }
        else { *lo = aux0; *n = 28; }
    }
};

struct Chunk {
    std::vector<uint32_t> prog;   // allocated [n][4]
    uint32_t n_regs = 1, alpha_off = 0;
};

struct ZcPlan {                      // everything that depends on a chip's program only (cached per process)
    uint32_t n_instr = 0, main_w = 0, prep_w = 0;
    bool macros_enabled = true;      // SP1HIP_ZC_MACRO when the plan was made (part of the cache key)
    bool mul_enabled = true;         // whether the MulOperation hints are honoured (the chip's height, see zc_get_plan)
    std::vector<uint32_t> source;    // the caller's [n][3] program (collision check)
    std::vector<uint32_t> prog;      // allocated [n][4], whole program (padded-row evaluation)
    uint32_t n_regs = 1;
    std::vector<Chunk> chunks, mono, fine;
    std::vector<uint32_t> sched;     // the scheduled SSA the forms above were cut from
    std::vector<ZcMacro> macros;     // hinted sub-AIRs evaluated by fused kernels (zc_poseidon2.hpp); their asserts are not in the forms above
    std::vector<ZcPoly> polys;       // the polynomial identities among them (zc_poly.hpp), by ZcMacro::aux0
    std::vector<std::vector<ZcPolySeg>> poly_segs;   // their device-table segments, ready for a proof's alpha
};

struct ChipState {
    const sp1hip_zc_chip_t* in;
    std::vector<ZcMacro> macros;
    std::shared_ptr<const ZcPlan> plan;   // (the polynomial identities' forms live in the plan)
    std::vector<size_t> poly_off;   // per macro: word offset of its device table in the call's constant blob (kind 7 only)
    const uint32_t* p_blob = nullptr;
    std::vector<uint32_t> prog;     // allocated [n][4]
    uint32_t n_regs = 1;
    std::vector<Ext> alpha_pows, gkr_pows;
    std::vector<Chunk> chunks;         // split at assert boundaries (parallel across constraints: the small rounds)
    std::vector<uint32_t> chunk_off;   // offset (in instructions) of each chunk inside d_prog
    std::vector<Chunk> mono;           // the undivided program (+ a TOUCH chunk): no recomputation (the large rounds)
    std::vector<uint32_t> mono_off;
    std::vector<Chunk> fine;           // short pieces for the last, latency-bound rounds
    std::vector<uint32_t> fine_off;
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
};

// linear-scan register allocation of the SSA program (host)
static inline bool zc_is_imm(uint32_t op) { return op >= ZC_ADDC && op <= ZC_MULC; }

// ADD / SUB / MUL with a CONST operand -> the immediate forms (same instruction indices; the CONST instructions stay
// behind and drop out when the chunks collect the cones of the asserts).
static void fold_immediates(const uint32_t* ssa, uint32_t n, std::vector<uint32_t>* out) {
    out->assign(ssa, ssa + (size_t)n * 3);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        if (op != ZC_ADD && op != ZC_SUB && op != ZC_MUL) continue;
        const bool ca = ssa[3 * a] == ZC_CONST, cb = ssa[3 * b] == ZC_CONST;
        if (ca == cb) continue;                                 // none (or both: left to the generic path)
        uint32_t* o = out->data() + 3 * (size_t)k;
        const uint32_t var = ca ? b : a, c = ssa[3 * (ca ? a : b) + 1];
        o[1] = var; o[2] = c;
        o[0] = op == ZC_ADD ? ZC_ADDC : op == ZC_MUL ? ZC_MULC : (cb ? ZC_SUBC : ZC_CSUB);
    }
}

// Instruction scheduling (host). The k-th ASSERT_ZERO of the caller's program is constraint k; here every assert gets
// its index as an explicit operand, which frees the ORDER: asserts are sorted by the last (or first) trace column their
// cone touches, and every value is emitted right before its first use (depth-first from the asserts), the columns an
// assert needs first, in ascending order (so that runs of them merge into one load instruction). Constraints of real
// chips are local in the column layout (an operation's columns are contiguous), so this keeps few values alive at a
// time: the register file of a 250-column chip shrinks from "every shared sub-expression of the chip" to the handful
// one operation needs, which is what decides the workgroup width / occupancy of the interpreter (launch_round).
// mode 0: original order (asserts tagged only); 1: by last column; 2: by first column; 3: by last column over a program whose
// cheap values are rematerialised at every use (below), each load emitted right before the instruction that reads it.

// Rematerialisation (host), for the FieldOpCols chips (round 5: secp256k1 add / double, uint256): their constraints are
// coefficient-wise convolutions sum_i a[i] b[k - i] over 32-limb operands that are COLUMNS, 63 coefficients per field operation,
// ten operations per row. With every column loaded once and kept, ~100 values are live throughout (two operands, the carry, the
// byte decompositions of the point) and a wave's register file takes 120 KB of LDS: ONE wave per compute unit. Here every use of
// a column (and of a value computed from columns in at most 4 instructions — the high byte (u16 - low) / 256 of a memory limb —
// or in at most 7 if it is used at most 8 times) gets its own copy right before the user; a product then costs LOAD, LOAD (forwarded), MAD
// instead of MAD, the program is ~3x longer — and the file shrinks to the accumulators and the few values that are worth keeping.
static void rematerialize_cheap(const uint32_t* ssa, uint32_t n, std::vector<uint32_t>* out) {
    auto is_bin = [](uint32_t op) { return op == ZC_ADD || op == ZC_SUB || op == ZC_MUL; };
    auto is_un = [](uint32_t op) { return op == ZC_NEG || op == ZC_ASSERT_ZERO || zc_is_imm(op); };
    std::vector<uint32_t> uses(n, 0), cone(n, 1);
    std::vector<char> remat(n, 0);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k];
        if (is_bin(op)) { uses[ssa[3 * k + 1]]++; uses[ssa[3 * k + 2]]++; }
        else if (is_un(op)) uses[ssa[3 * k + 1]]++;
    }
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        if (op == ZC_ASSERT_ZERO) continue;
        if (is_bin(op)) { cone[k] = cone[a] + cone[b] + 1; remat[k] = remat[a] && remat[b] && (cone[k] <= 4 || (cone[k] <= 7 && uses[k] <= 8)); }
        else if (is_un(op)) { cone[k] = cone[a] + 1; remat[k] = remat[a] && (cone[k] <= 4 || (cone[k] <= 7 && uses[k] <= 8)); }
        else remat[k] = 1;                       // LOAD_MAIN / LOAD_PREP / CONST / PUBLIC
    }
    out->clear();
    std::vector<uint32_t> where(n, 0xffffffffu), stack;
    auto push = [&](uint32_t op, uint32_t a, uint32_t b) { out->insert(out->end(), {op, a, b}); return (uint32_t)(out->size() / 3 - 1); };
    // a fresh copy of the cone of a rematerialisable value (at most 7 instructions: recursion depth is bounded)
    std::function<uint32_t(uint32_t)> clone = [&](uint32_t v) -> uint32_t {
        if (!remat[v]) return where[v];
        const uint32_t op = ssa[3 * v], a = ssa[3 * v + 1], b = ssa[3 * v + 2];
        if (is_bin(op)) { const uint32_t x = clone(a), y = clone(b); return push(op, x, y); }
        if (is_un(op)) { const uint32_t x = clone(a); return push(op, x, b); }
        return push(op, a, b);
    };
    for (uint32_t k = 0; k < n; k++) {
        if (remat[k]) continue;                  // emitted where it is used
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        if (is_bin(op)) { const uint32_t x = clone(a), y = clone(b); where[k] = push(op, x, y); }
        else if (is_un(op)) { const uint32_t x = clone(a); where[k] = push(op, x, b); }
        else where[k] = push(op, a, b);
    }
}

static void schedule_program(const uint32_t* ssa, uint32_t n, uint32_t main_w, int mode, std::vector<uint32_t>* out) {
    std::vector<uint32_t> remat_ssa;
    const bool lazy = mode == 3;
    if (lazy) {
        rematerialize_cheap(ssa, n, &remat_ssa);
        ssa = remat_ssa.data(); n = (uint32_t)(remat_ssa.size() / 3); mode = 1;
    }
    auto is_bin = [](uint32_t op) { return op == ZC_ADD || op == ZC_SUB || op == ZC_MUL; };
    auto is_un = [](uint32_t op) { return op == ZC_NEG || op == ZC_ASSERT_ZERO || zc_is_imm(op); };
    std::vector<uint32_t> asserts, idx_of(n, 0);
    for (uint32_t k = 0; k < n; k++)
        if (ssa[3 * k] == ZC_ASSERT_ZERO) { idx_of[k] = (uint32_t)asserts.size(); asserts.push_back(k); }
    out->clear();
    if (mode == 0) {
        out->assign(ssa, ssa + (size_t)n * 3);
        for (uint32_t k : asserts) (*out)[3 * (size_t)k + 2] = idx_of[k];
        return;
    }
    // first / last column a value depends on (main columns first, then preprocessed)
    std::vector<uint32_t> lo(n, 0xffffffffu), hi(n, 0);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        if (op == ZC_LOAD_MAIN) lo[k] = hi[k] = a + 1;
        else if (op == ZC_LOAD_PREP) lo[k] = hi[k] = main_w + a + 1;
        else if (is_bin(op)) { lo[k] = std::min(lo[a], lo[b]); hi[k] = std::max(hi[a], hi[b]); }
        else if (is_un(op)) { lo[k] = lo[a]; hi[k] = hi[a]; }
    }
    std::vector<uint32_t> order = asserts;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return mode == 1 ? hi[x] < hi[y] : lo[x] < lo[y]; });
    std::vector<uint32_t> renum(n, 0xffffffffu), stack, loads;
    auto emit = [&](uint32_t k) {
        const uint32_t op = ssa[3 * k];
        uint32_t a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        if (is_bin(op)) { a = renum[a]; b = renum[b]; }
        else if (is_un(op)) a = renum[a];
        if (op == ZC_ASSERT_ZERO) b = idx_of[k];
        renum[k] = (uint32_t)(out->size() / 3);
        out->insert(out->end(), {op, a, b});
    };
    std::vector<uint8_t> visited(n, 0);
    for (uint32_t as : order) {
        // 1. the columns this assert still needs, ascending
        loads.clear();
        stack.assign(1, ssa[3 * as + 1]);
        std::vector<uint32_t> seen_here;
        while (!stack.empty()) {
            const uint32_t v = stack.back();
            stack.pop_back();
            if (renum[v] != 0xffffffffu || visited[v]) continue;
            visited[v] = 1;
            seen_here.push_back(v);
            const uint32_t op = ssa[3 * v];
            if (op == ZC_LOAD_MAIN || op == ZC_LOAD_PREP) loads.push_back(v);
            else if (is_bin(op)) { stack.push_back(ssa[3 * v + 1]); stack.push_back(ssa[3 * v + 2]); }
            else if (is_un(op)) stack.push_back(ssa[3 * v + 1]);
        }
        for (uint32_t v : seen_here) visited[v] = 0;
        std::sort(loads.begin(), loads.end(), [&](uint32_t x, uint32_t y) { return lo[x] < lo[y]; });
        if (!lazy) for (uint32_t v : loads) emit(v);
        // 2. the rest of the cone, operands before users (iterative post-order)
        stack.assign(1, ssa[3 * as + 1]);
        while (!stack.empty()) {
            const uint32_t v = stack.back();
            if (renum[v] != 0xffffffffu) { stack.pop_back(); continue; }
            const uint32_t op = ssa[3 * v];
            uint32_t need[2], nn = 0;
            if (is_bin(op)) { need[nn++] = ssa[3 * v + 1]; need[nn++] = ssa[3 * v + 2]; }
            else if (is_un(op)) need[nn++] = ssa[3 * v + 1];
            bool ready = true;
            for (uint32_t j = nn; j-- > 0;)
                if (renum[need[j]] == 0xffffffffu) { stack.push_back(need[j]); ready = false; }
            if (ready) { emit(v); stack.pop_back(); }
        }
        emit(as);
    }
}

// Register allocation of an SSA program (host) -> the interpreter's [op | flags, dst, a, b] words.
//  * last-use allocation into the lowest free register (the LDS file is sized by the highest one used);
//  * operand forwarding: the interpreter keeps the value of the last value-producing instruction in VGPRs (`prev`);
//    an operand that is that value is flagged ZC_A_PREV / ZC_B_PREV (no LDS read), and a value whose every use
//    happens before the next value is produced is flagged ZC_DST_TEMP: it never touches the register file — in the
//    constraint programs of real chips about half of all values are consumed by the very next instruction;
//  * runs of up to 4 LOADs of consecutive columns of one table become ONE instruction (count in bits 16-17) with
//    consecutive destination registers: the kernel issues all their global loads before waiting once, so a row
//    of a wide chip costs a quarter of the memory round trips.
static int allocate_registers(const uint32_t* ssa, uint32_t n, std::vector<uint32_t>* out, uint32_t* n_regs) {
    auto is_bin = [](uint32_t op) { return op == ZC_ADD || op == ZC_SUB || op == ZC_MUL; };
    auto is_un = [](uint32_t op) { return op == ZC_NEG || op == ZC_ASSERT_ZERO || zc_is_imm(op); };
    std::vector<int> last_use(n, -1), next_val(n, -1);
    std::vector<uint32_t> n_uses(n, 0);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        SP1HIP_REQUIRE(op <= ZC_ASSERT_ZERO || zc_is_imm(op), "bad opcode in constraint program");
        if (is_bin(op)) {
            SP1HIP_REQUIRE(a < k && b < k, "constraint program is not in SSA order");
            last_use[a] = (int)k; last_use[b] = (int)k;
            n_uses[a]++; n_uses[b]++;
        } else if (is_un(op)) {
            SP1HIP_REQUIRE(a < k, "constraint program is not in SSA order");
            last_use[a] = (int)k;
            n_uses[a]++;
        }
    }
    // fused[k]: a MULC whose single use is the ADD / SUB (as subtrahend) that is the next value-producing instruction:
    // it is not emitted, its user becomes a MADC
    std::vector<char> fused(n, 0);
    std::vector<int> fused_src(n, -1);         // for the user: the MULC it absorbs
    static const bool madc_enabled = [] { const char* e = getenv("SP1HIP_ZC_MADC"); return !(e && e[0] == '0'); }();
    static const bool mad_enabled = [] { const char* e = getenv("SP1HIP_ZC_MAD"); return !(e && e[0] == '0'); }();
    if (madc_enabled || mad_enabled)
        for (uint32_t k = 0; k + 1 < n; k++) {
            const bool is_mulc = ssa[3 * k] == ZC_MULC && madc_enabled;
            const bool is_mul = ssa[3 * k] == ZC_MUL && mad_enabled && ssa[3 * k + 1] != ssa[3 * k + 2];
            if (!(is_mulc || is_mul) || n_uses[k] != 1) continue;
            uint32_t u = k + 1;
            while (u < n && ssa[3 * u] == ZC_ASSERT_ZERO) u++;
            if (u >= n || fused_src[u] >= 0) continue;
            const uint32_t uop = ssa[3 * u], ua = ssa[3 * u + 1], ub = ssa[3 * u + 2];
            if (ua == ub) continue;
            if (is_mul) {                      // the other summand must not be one of the factors (it may live in `prev` only)
                const uint32_t other = ua == k ? ub : ua;
                if (other == ssa[3 * k + 1] || other == ssa[3 * k + 2]) continue;
            }
            if ((uop == ZC_ADD && (ua == k || ub == k)) || (uop == ZC_SUB && ub == k)) { fused[k] = 1; fused_src[u] = (int)k; }
        }
    {   // next_val[k]: the first value-producing (emitted) instruction after k
        int nv = -1;
        for (uint32_t k = n; k-- > 0;) { next_val[k] = nv; if (ssa[3 * k] != ZC_ASSERT_ZERO && !fused[k]) nv = (int)k; }
    }
    // load groups: group_len[k] > 0 on the first LOAD of a run, 0 on the merged followers
    std::vector<uint32_t> group_len(n, 1);
    for (uint32_t k = 0; k < n;) {
        const uint32_t op = ssa[3 * k];
        uint32_t m = 1;
        if (op == ZC_LOAD_MAIN || op == ZC_LOAD_PREP)
            while (m < 4 && k + m < n && ssa[3 * (k + m)] == op && ssa[3 * (k + m) + 1] == ssa[3 * k + 1] + m) m++;
        group_len[k] = m;
        for (uint32_t j = 1; j < m; j++) group_len[k + j] = 0;
        k += m;
    }
    // a value is a temporary when it dies before the next value is produced (and it is not inside a load group,
    // whose members all go to the file except that the LAST column stays forwardable)
    auto is_temp = [&](uint32_t k) {
        if (ssa[3 * k] == ZC_ASSERT_ZERO || n_uses[k] == 0) return false;
        if ((ssa[3 * k] == ZC_LOAD_MAIN || ssa[3 * k] == ZC_LOAD_PREP) && !(group_len[k] == 1)) return false;
        return next_val[k] < 0 ? true : last_use[k] <= next_val[k];
    };
    std::vector<char> busy;
    auto take = [&](uint32_t m) {            // lowest run of m free registers
        uint32_t run = 0;
        for (uint32_t r = 0; r < busy.size(); r++) {
            run = busy[r] ? 0 : run + 1;
            if (run == m) { for (uint32_t j = 0; j < m; j++) busy[r - j] = 1; return r + 1 - m; }
        }
        const uint32_t tail = run;             // free registers at the top can be extended
        const uint32_t start = (uint32_t)busy.size() - tail;
        busy.resize(start + m, 1);
        for (uint32_t j = 0; j < m; j++) busy[start + j] = 1;
        return start;
    };
    std::vector<uint32_t> reg_of(n, 0xffffffffu);
    out->clear();
    int last_value = -1;                       // SSA index held in `prev` when the next instruction runs
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        if (group_len[k] == 0) continue;       // merged into the group's first LOAD
        if (fused[k]) continue;                // emitted with its user
        uint32_t word = op, ra = a, rb = b;
        if (fused_src[k] >= 0 && ssa[3 * fused_src[k]] == ZC_MUL) {      // acc +- (x * y)  ->  MAD / MSB
            const uint32_t m = (uint32_t)fused_src[k], acc = a == m ? b : a;
            uint32_t fx = ssa[3 * m + 1], fy = ssa[3 * m + 2];
            uint32_t racc = 0, rx = 0, ry = 0;
            word = op == ZC_SUB ? ZC_MSB : ZC_MAD;
            if ((int)acc == last_value) word |= ZC_B_PREV;
            else {
                racc = reg_of[acc];
                if ((int)fy == last_value) std::swap(fx, fy);        // the forwarded factor must be the first one
                if ((int)fx == last_value) word |= ZC_A_PREV;
            }
            if (!(word & ZC_A_PREV)) rx = reg_of[fx];
            ry = reg_of[fy];
            if ((!(word & ZC_A_PREV) && rx == 0xffffffffu) || ry == 0xffffffffu || (!(word & ZC_B_PREV) && racc == 0xffffffffu)) {
                set_error("internal: operand of fused multiply-add %u has no register", k);
                return SP1HIP_ERROR_RUNTIME;
            }
            if (word & ZC_A_PREV) rx = 0;
            for (uint32_t f : {fx, fy})
                if (last_use[f] == (int)m && reg_of[f] != 0xffffffffu) { busy[reg_of[f]] = 0; reg_of[f] = 0xffffffffu; }
            if (last_use[acc] == (int)k && reg_of[acc] != 0xffffffffu) { busy[reg_of[acc]] = 0; reg_of[acc] = 0xffffffffu; }
            uint32_t dst = 0;
            if (is_temp(k) || n_uses[k] == 0) word |= ZC_DST_TEMP;
            else { dst = take(1); reg_of[k] = dst; }
            if (dst > 0xffffu || racc > 0xffffu) { set_error("constraint program needs more than 65536 registers"); return SP1HIP_ERROR_RUNTIME; }
            last_value = (int)k;
            out->insert(out->end(), {word, dst | (racc << 16), rx, ry});
            continue;
        }
        if (fused_src[k] >= 0) {               // acc +- (term * c)  ->  MADC
            const uint32_t m = (uint32_t)fused_src[k], term = ssa[3 * m + 1], acc = a == m ? b : a;
            const uint32_t c = kb::to_monty(ssa[3 * m + 2] % kb::P);
            uint32_t racc = 0;
            word = ZC_MADC;
            if ((int)term == last_value) word |= ZC_A_PREV; else ra = reg_of[term];
            if ((int)acc == last_value) word |= ZC_B_PREV; else racc = reg_of[acc];
            if ((!(word & ZC_A_PREV) && ra == 0xffffffffu) || (!(word & ZC_B_PREV) && racc == 0xffffffffu)) {
                set_error("internal: operand of fused instruction %u has no register", k);
                return SP1HIP_ERROR_RUNTIME;
            }
            if (word & ZC_A_PREV) ra = 0;
            if (last_use[term] == (int)m && reg_of[term] != 0xffffffffu) { busy[reg_of[term]] = 0; reg_of[term] = 0xffffffffu; }
            if (acc != term && last_use[acc] == (int)k && reg_of[acc] != 0xffffffffu) { busy[reg_of[acc]] = 0; reg_of[acc] = 0xffffffffu; }
            uint32_t dst = 0;
            if (is_temp(k) || n_uses[k] == 0) word |= ZC_DST_TEMP;
            else { dst = take(1); reg_of[k] = dst; }
            if (dst > 0xffffu || racc > 0xffffu) { set_error("constraint program needs more than 65536 registers"); return SP1HIP_ERROR_RUNTIME; }
            last_value = (int)k;
            out->insert(out->end(), {word, dst | (racc << 16), ra, op == ZC_SUB ? kb::neg(c) : c});
            continue;
        }
        if (is_bin(op) || is_un(op)) {
            // the interpreter loads operand A into the forwarded value's registers: a forwarded operand must BE operand A
            uint32_t oa = a, ob = b;
            if (is_bin(op) && (int)ob == last_value && (int)oa != last_value) {
                std::swap(oa, ob);
                if (op == ZC_SUB) word = ZC_RSUB;
            }
            ra = oa; rb = ob;
            if ((int)oa == last_value) word |= ZC_A_PREV; else ra = reg_of[oa];
            if (is_bin(op)) { if ((int)ob == last_value) word |= ZC_B_PREV; else rb = reg_of[ob]; }
            if ((!(word & ZC_A_PREV) && ra == 0xffffffffu) || (is_bin(op) && !(word & ZC_B_PREV) && rb == 0xffffffffu)) {
                set_error("internal: operand of instruction %u has no register", k);
                return SP1HIP_ERROR_RUNTIME;
            }
            if (last_use[a] == (int)k && reg_of[a] != 0xffffffffu) { busy[reg_of[a]] = 0; reg_of[a] = 0xffffffffu; }
            if (is_bin(op) && b != a && last_use[b] == (int)k && reg_of[b] != 0xffffffffu) { busy[reg_of[b]] = 0; reg_of[b] = 0xffffffffu; }
        }
        uint32_t dst = 0;
        if (op != ZC_ASSERT_ZERO) {
            const uint32_t m = group_len[k];
            if (m == 1 && (is_temp(k) || n_uses[k] == 0)) {
                word |= ZC_DST_TEMP;           // lives in `prev` only (or is dead)
            } else {
                dst = take(m);
                for (uint32_t j = 0; j < m; j++) {
                    if (n_uses[k + j]) reg_of[k + j] = dst + j; else busy[dst + j] = 0;
                }
            }
            word |= (m - 1) << 16;
            last_value = (int)(k + m - 1);
        }
        if (op == ZC_CONST) ra = kb::to_monty(a % kb::P);
        if (zc_is_imm(op)) rb = kb::to_monty(b % kb::P);
        if (op == ZC_ASSERT_ZERO) rb = b;         // the constraint's index (schedule_program)
        out->insert(out->end(), {word, dst, ra, rb});
    }
    *n_regs = busy.empty() ? 1u : (uint32_t)busy.size();
    return SP1HIP_SUCCESS;
}

// Splits the SSA program into self-contained chunks at assert boundaries (each chunk re-emits the
// dependency cone of its asserts, at most ~`limit` instructions unless a single cone is larger). Chunks
// are independent workgroups on the GPU: wide chips get parallelism across constraints, which is what
// keeps the late, tiny sumcheck rounds from being one wave interpreting thousands of instructions
// serially (cf. the reference's chunked bytecode, /root/reference/sp1-gpu/crates/air/src/ir/bytecode.rs:L27-L110).
static int build_chunks(const uint32_t* ssa, uint32_t n, uint32_t main_w, uint32_t prep_w, uint32_t limit,
                        std::vector<Chunk>* out, uint32_t hard_max = ZC_CHUNK_HARD_MAX, const std::vector<ZcMacro>* macros = nullptr,
                        const std::vector<ZcPoly>* polys = nullptr) {
    std::vector<uint32_t> stamp(n, 0xffffffffu);
    std::vector<uint8_t> cone_seen(n, 0);
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
            else if (op == ZC_NEG || op == ZC_ASSERT_ZERO || zc_is_imm(op)) a = renum[a];
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
            else if (op == ZC_NEG || zc_is_imm(op)) stack.push_back(ssa[3 * v + 1]);
        }
        if (!asserts.empty() && members.size() + fresh.size() + asserts.size() + 1 > limit) {
            // over the target size. If most of this assert's cone is ALREADY in the chunk (it shares the chunk's
            // intermediate values: the 16 constraints of a Poseidon2 external round share one S-box / linear layer), closing
            // the chunk here would recompute all of it in the next one: keep it, up to a hard cap.
            bool keep = false;
            if (limit != 0xffffffffu && members.size() + fresh.size() + asserts.size() + 1 <= hard_max) {
                size_t cone = 0;
                std::vector<uint32_t> st2(1, ssa[3 * k + 1]);
                std::vector<uint8_t>& seen = cone_seen;
                std::vector<uint32_t> touched;
                while (!st2.empty()) {
                    const uint32_t v = st2.back();
                    st2.pop_back();
                    if (seen[v]) continue;
                    seen[v] = 1; touched.push_back(v); cone++;
                    const uint32_t op = ssa[3 * v];
                    if (op == ZC_ADD || op == ZC_SUB || op == ZC_MUL) { st2.push_back(ssa[3 * v + 1]); st2.push_back(ssa[3 * v + 2]); }
                    else if (op == ZC_NEG || zc_is_imm(op)) st2.push_back(ssa[3 * v + 1]);
                }
                for (uint32_t v : touched) seen[v] = 0;
                keep = 2 * fresh.size() <= cone;
            }
            if (!keep) {
                for (uint32_t v : fresh) stamp[v] = 0xffffffffu;     // undo, close the chunk, retry in a new one
                SP1HIP_TRY(flush());
                k--;
                continue;
            }
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
    if (macros)                                    // the fused pieces of a hinted sub-AIR carry the GKR term of its columns themselves
        for (const ZcMacro& m : *macros) {
            uint32_t lo, cnt;
            m.owned(&lo, &cnt);
            for (uint32_t c = 0; c < cnt; c++) seen_m[lo + c] = true;
            if (m.kind == ZC_HINT_POLY && polys) for (uint32_t c : (*polys)[m.aux0].owned) seen_m[c] = true;
        }
    for (auto& c : *out)
        for (size_t k = 0; k < c.prog.size() / 4; k++) {
            uint32_t* o = c.prog.data() + 4 * k;
            const uint32_t op = o[0] & 0xffu, cnt = ((o[0] >> 16) & 3u) + 1;
            if (op != ZC_LOAD_MAIN && op != ZC_LOAD_PREP) continue;
            std::vector<bool>& seen = op == ZC_LOAD_MAIN ? seen_m : seen_p;
            for (uint32_t j = 0; j < cnt; j++)
                if (!seen[o[2] + j]) { seen[o[2] + j] = true; o[0] |= ZC_GKR_FLAG << j; }
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

// Host interpreter of allocated program words on ONE row (Montgomery words; null row = all zeros): every ASSERT_ZERO hands
// (constraint index, value) to `on_assert`. The same semantics as run_program on the device, in the base field.
template <class F>
static void eval_words_row(const uint32_t* words, size_t n, uint32_t n_regs, const uint32_t* main_row, const uint32_t* prep_row,
                           const uint32_t* publics, F&& on_assert) {
    std::vector<uint32_t> reg(n_regs + 4, 0);
    uint32_t prev = 0;
    for (size_t k = 0; k < n; k++) {
        const uint32_t opw = words[4 * k], op = opw & 0xffu, dst = words[4 * k + 1], x = words[4 * k + 2], y = words[4 * k + 3];
        const uint32_t A = (opw & ZC_A_PREV) ? prev : (op >= ZC_ADD && op != ZC_TOUCH ? reg[x] : 0u);
        const bool bin = (op >= ZC_ADD && op <= ZC_MUL) || op == ZC_RSUB;
        const uint32_t B = (bin && (opw & ZC_B_PREV)) ? prev : (bin ? reg[y] : 0u);
        uint32_t res = 0;
        switch (op) {
            case ZC_LOAD_MAIN: case ZC_LOAD_PREP: {
                const uint32_t* row = op == ZC_LOAD_MAIN ? main_row : prep_row;
                for (uint32_t j = 0; j <= ((opw >> 16) & 3u); j++) {
                    prev = row ? row[x + j] : 0u;
                    if (!(opw & ZC_DST_TEMP)) reg[dst + j] = prev;
                }
                continue;
            }
            case ZC_TOUCH: continue;
            case ZC_CONST: res = x; break;
            case ZC_PUBLIC: res = publics[x]; break;
            case ZC_ADD: res = kb::add(A, B); break;
            case ZC_SUB: res = kb::sub(A, B); break;
            case ZC_MUL: res = kb::mul(A, B); break;
            case ZC_RSUB: res = kb::sub(B, A); break;
            case ZC_NEG: res = kb::neg(A); break;
            case ZC_ADDC: res = kb::add(A, y); break;
            case ZC_SUBC: res = kb::sub(A, y); break;
            case ZC_CSUB: res = kb::sub(y, A); break;
            case ZC_MULC: res = kb::mul(A, y); break;
            case ZC_MADC: res = kb::add((opw & ZC_B_PREV) ? prev : reg[dst >> 16], kb::mul(A, y)); break;
            case ZC_MAD: res = kb::add((opw & ZC_B_PREV) ? prev : reg[dst >> 16], kb::mul(A, reg[y])); break;
            case ZC_MSB: res = kb::sub((opw & ZC_B_PREV) ? prev : reg[dst >> 16], kb::mul(A, reg[y])); break;
            default: on_assert(y, A); continue;
        }
        prev = res;
        if (!(opw & ZC_DST_TEMP)) reg[dst & 0xffffu] = res;
    }
}

// host evaluation of the program on an all-zero row (padded_row_adjustment, shard.rs:L524-L536)
static Ext eval_zero_row(const ChipState& c, const uint32_t* publics) {
    Ext acc = kb::ext_zero();
    eval_words_row(c.prog.data(), c.prog.size() / 4, c.n_regs, nullptr, nullptr, publics,
                   [&](uint32_t idx, uint32_t v) { acc = acc + kb::ext_mul_base(c.alpha_pows[idx], v); });
    return acc;
}

// host model of the fused pieces on ONE row of base-field words (the planner's check of a hint, sp1hip_zerocheck_plan_eval)
template <class Sink>
static void macro_eval_row(const ZcMacro& m, const std::vector<ZcPoly>& polys, const uint32_t* main_row, Sink&& sink) {
    static const p2::RoundConstants host_rc = p2::make_round_constants();
    if (m.kind == ZC_HINT_POLY) { zc_poly_eval_row(polys[m.aux0], main_row, sink); return; }
    for (uint32_t q = 0; q < m.n_host_pieces(); q++) {
        if (m.kind == ZC_HINT_POSEIDON2)
            zc_p2_piece<P2Base>(q, &host_rc, [&](uint32_t c, bool) { return main_row[m.base_col + c]; }, sink);
        else if (m.kind == ZC_HINT_KECCAK)
            zc_keccak_piece<P2Base>(q, [&](uint32_t c, bool) { return main_row[m.base_col + c]; }, sink);
        else if (m.kind == ZC_HINT_MUL)
            zc_mul_piece<P2Base>(q, [&](uint32_t c, bool) { return main_row[m.base_col + c]; }, [&](uint32_t c, bool) { return main_row[m.aux0 + c]; }, sink);
        else if (m.kind == ZC_HINT_SEPTIC_CURVE)
            zc_septic_curve_piece<P2Base>([&](uint32_t c, bool) { return main_row[m.base_col + c]; }, sink);
        else
            zc_septic_sum_piece<P2Base>(q, [&](uint32_t c, bool) { return main_row[m.base_col + c]; },
                                        [&](uint32_t c, bool) { return main_row[m.aux0 + c]; }, [&]() { return main_row[m.aux1]; }, sink);
    }
}

static uint32_t asserts_total(const uint32_t* program, uint32_t n) {
    uint32_t a = 0;
    for (uint32_t k = 0; k < n; k++) a += program[3 * k] == ZC_ASSERT_ZERO;
    return a;
}

// The plan of a program (immediates folded, instruction order chosen, registers allocated; chunked, undivided and finely
// cut forms) depends on the program alone: a machine's chips are planned once per process and looked up afterwards (a
// prover proves the same machine shard after shard; planning 33 chips costs ~1.3 ms of host time per proof).
// `rows`: the chip's height in this proof. The MulOperation piece (kind 6) replaces interpreter work that grows with the height by
// one more launch per round: below ZC_MUL_MIN_ROWS rows (SP1HIP_ZC_MUL_MIN_ROWS) that launch sits at its latency floor in every
// round and the hint is ignored — the recorded core shard has 128 Mul rows, a fibonacci shard 1.9 million.
constexpr uint64_t ZC_MUL_MIN_ROWS = 1u << 16;
static int zc_get_plan(const uint32_t* program, uint32_t n_instr, uint32_t main_width, uint32_t prep_width, int chip_index,
                       std::shared_ptr<const ZcPlan>* out, uint64_t rows = ~0ull) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint32_t v) { h = (h ^ v) * 1099511628211ull; };
    // SP1HIP_ZC_MACRO=0 ignores hints; read per call like the BIVARIATE / FORK switches and part of the cache key
    const bool macros_enabled = [] { const char* e = getenv("SP1HIP_ZC_MACRO"); return !(e && e[0] == '0'); }();
    const uint64_t mul_min_rows = [] { const char* e = getenv("SP1HIP_ZC_MUL_MIN_ROWS"); return e ? (uint64_t)strtoull(e, nullptr, 10) : ZC_MUL_MIN_ROWS; }();
    const bool mul_enabled = macros_enabled && rows >= mul_min_rows;
    mix(main_width); mix(prep_width); mix(n_instr); mix((macros_enabled ? 1u : 0u) | (mul_enabled ? 2u : 0u));
    for (size_t k = 0; k < (size_t)n_instr * 3; k++) mix(program[k]);
    static std::mutex plan_mutex;
    static std::unordered_map<uint64_t, std::shared_ptr<const ZcPlan>> plan_cache;
    std::shared_ptr<const ZcPlan> plan;
    {
        std::lock_guard<std::mutex> lk(plan_mutex);
        auto it = plan_cache.find(h);
        if (it != plan_cache.end() && it->second->n_instr == n_instr && it->second->main_w == main_width && it->second->prep_w == prep_width &&
            it->second->macros_enabled == macros_enabled && it->second->mul_enabled == mul_enabled && (n_instr == 0 || memcmp(it->second->source.data(), program, (size_t)n_instr * 12) == 0))
            plan = it->second;
    }
    if (!plan) {
        std::shared_ptr<ZcPlan> np(new ZcPlan());
        np->n_instr = n_instr; np->main_w = main_width; np->prep_w = prep_width; np->macros_enabled = macros_enabled; np->mul_enabled = mul_enabled;
        np->source.assign(program, program + (size_t)n_instr * 3);
        // hinted sub-AIRs (zc_poseidon2.hpp): the HINT pseudo-instructions become harmless constants, the hints are CHECKED
        // against the SSA, and the asserts they cover leave the interpreted forms (not the whole program `prog`, which the
        // host still evaluates on the all-zero row)
        std::vector<uint32_t> clean(program, program + (size_t)n_instr * 3);
        {
            uint32_t asserts_before = 0;
            for (uint32_t k = 0; k < n_instr; k++) {
                if (clean[3 * k] == ZC_ASSERT_ZERO) asserts_before++;
                if (clean[3 * k] != ZC_HINT) continue;
                const uint32_t kind = clean[3 * k + 1] & 0xffu, w1 = clean[3 * k + 1] >> 8, w2 = clean[3 * k + 2];
                if (kind == ZC_HINT_POLY) {
                    // a polynomial identity (zc_poly.hpp): the values it names follow as ARG pseudo-instructions. Its forms are taken
                    // from the SSA; values that are not affine in the main columns drop the hint (the interpreter keeps the constraints)
                    const uint32_t n_terms = w1, n_c = w2;
                    SP1HIP_REQUIRE(n_terms <= 80 && n_c >= 1 && n_c < (1u << 16), "polynomial-identity hint: bad header");
                    std::vector<std::vector<uint32_t>> ids(3 * (size_t)n_terms + 1);
                    uint32_t j = k + 1;
                    for (; j < n_instr && clean[3 * j] == ZC_HINT && (clean[3 * j + 1] & 0xffu) == ZC_HINT_POLY_ARG; j++) {
                        const uint32_t code = clean[3 * j + 1] >> 8, id = clean[3 * j + 2];
                        SP1HIP_REQUIRE((code == 255u || code < 3 * n_terms) && id < k, "polynomial-identity hint: bad argument");
                        ids[code == 255u ? 3 * (size_t)n_terms : code].push_back(id);
                    }
                    bool shape = ids.back().size() == n_c;
                    for (uint32_t t = 0; t < n_terms; t++)
                        shape &= !ids[3 * t].empty() && !ids[3 * t + 1].empty() && ids[3 * t].size() + ids[3 * t + 1].size() + std::max<size_t>(ids[3 * t + 2].size(), 1) - 2 <= n_c;
                    SP1HIP_REQUIRE(shape, "polynomial-identity hint: operand counts do not match the number of constraints");
                    for (uint32_t q = k; q < j; q++) { clean[3 * q] = ZC_CONST; clean[3 * q + 1] = 0; clean[3 * q + 2] = 0; }
                    ZcPoly poly;
                    poly.first_constraint = asserts_before; poly.n_c = n_c;
                    bool affine = n_terms <= ZC_POLY_MAX_TERMS;
                    std::vector<uint32_t> all;
                    for (auto& v : ids) all.insert(all.end(), v.begin(), v.end());
                    std::vector<ZcLinForm> forms;
                    affine = affine && zc_poly_extract(clean.data(), n_instr, all, &forms);
                    if (affine) {
                        size_t at = 0;
                        poly.terms.resize(n_terms);
                        for (uint32_t t = 0; t < n_terms; t++)
                            for (int f = 0; f < 3; f++) {
                                poly.terms[t].f[f].assign(forms.begin() + at, forms.begin() + at + ids[3 * t + f].size());
                                at += ids[3 * t + f].size();
                            }
                        poly.rest.assign(forms.begin() + at, forms.end());
                        ZcMacro m{kind, 0u, asserts_before};
                        m.aux0 = (uint32_t)np->polys.size(); m.n_c = n_c;
                        np->polys.push_back(std::move(poly));
                        np->macros.push_back(m);
                    } else if (getenv("SP1HIP_ZC_DEBUG")) {
                        fprintf(stderr, "[sp1hip zc] chip %d: polynomial-identity hint at constraint %u dropped (a named value is not affine in the main columns)\n", chip_index, asserts_before);
                    }
                    k = j - 1;
                    continue;
                }
                SP1HIP_REQUIRE(kind != ZC_HINT_POLY_ARG, "polynomial-identity argument without its hint");
                SP1HIP_REQUIRE((kind >= ZC_HINT_POSEIDON2 && kind <= ZC_HINT_SEPTIC_SUM) || kind == ZC_HINT_KECCAK || kind == ZC_HINT_MUL, "unknown hint kind in constraint program");
                ZcMacro m{kind, kind == ZC_HINT_SEPTIC_SUM ? (w2 & 0xffffu) : w2, asserts_before};
                if (kind == ZC_HINT_SEPTIC_SUM) { m.aux0 = w2 >> 16; m.aux1 = w1; }
                if (kind == ZC_HINT_MUL) m.aux0 = w1;                 // the first limb of op_b's value (op_c's: seven columns further)
                SP1HIP_REQUIRE((uint64_t)m.base_col + (kind == ZC_HINT_POSEIDON2 ? ZC_P2_COLUMNS : kind == ZC_HINT_KECCAK ? KK_IS_REAL + 1 : kind == ZC_HINT_MUL ? MUL_COLUMNS : 14u) <= main_width &&
                               (kind != ZC_HINT_SEPTIC_SUM || ((uint64_t)m.aux0 + 28 <= main_width && m.aux1 < main_width)) &&
                               (kind != ZC_HINT_MUL || (uint64_t)m.aux0 + MUL_OPC_FROM_OPB + 4 <= main_width), "hint: columns out of range");
                np->macros.push_back(m);
                clean[3 * k] = ZC_CONST; clean[3 * k + 1] = 0; clean[3 * k + 2] = 0;
            }
            if (!macros_enabled) np->macros.clear();
            if (!mul_enabled) np->macros.erase(std::remove_if(np->macros.begin(), np->macros.end(), [](const ZcMacro& m) { return m.kind == ZC_HINT_MUL; }), np->macros.end());
            // each hint is checked against the SSA on its own (below); two hints that overlap — a duplicated HINT, two sum
            // checkers sharing accumulator columns — would each pass and then count their constraints and the GKR batching
            // term of their columns twice: a silently invalid proof. Constraint ranges and owned columns must be disjoint.
            for (size_t a = 0; a < np->macros.size(); a++)
                for (size_t b = a + 1; b < np->macros.size(); b++) {
                    const ZcMacro &ma = np->macros[a], &mb = np->macros[b];
                    const bool c_overlap = ma.first_constraint < mb.first_constraint + mb.n_constraints() &&
                                           mb.first_constraint < ma.first_constraint + ma.n_constraints();
                    uint32_t alo, an, blo, bn;
                    ma.owned(&alo, &an); mb.owned(&blo, &bn);
                    const bool o_overlap = alo < blo + bn && blo < alo + an;
                    SP1HIP_REQUIRE(!c_overlap, "two fused-kernel hints cover the same constraints");
                    SP1HIP_REQUIRE(!o_overlap, "two fused-kernel hints own the same columns");
                }
        }
        program = clean.data();
        auto hinted = [&](uint32_t idx) {
            for (const ZcMacro& m : np->macros) if (idx >= m.first_constraint && idx < m.first_constraint + m.n_constraints()) return true;
            return false;
        };
        // the columns a polynomial identity OWNS (its piece carries their GKR batching term, the interpreter never loads them): those
        // of its rest form that neither its products, nor another identity, nor any constraint left to the interpreter reads
        if (!np->polys.empty() && std::all_of(np->macros.begin(), np->macros.end(), [](const ZcMacro& m) { return m.kind == ZC_HINT_POLY; })) {
            std::vector<uint8_t> interp(main_width, 0), visited(n_instr, 0);
            std::vector<uint32_t> stack;
            uint32_t idx = 0;
            for (uint32_t k = 0; k < n_instr; k++) {
                if (clean[3 * k] != ZC_ASSERT_ZERO) continue;
                if (!hinted(idx) && clean[3 * k + 1] < n_instr) stack.push_back(clean[3 * k + 1]);
                idx++;
            }
            while (!stack.empty()) {
                const uint32_t v = stack.back();
                stack.pop_back();
                if (visited[v]) continue;
                visited[v] = 1;
                const uint32_t op = clean[3 * v], a = clean[3 * v + 1], b = clean[3 * v + 2];
                if (op == ZC_LOAD_MAIN) { if (a < main_width) interp[a] = 1; }
                else if (op == ZC_ADD || op == ZC_SUB || op == ZC_MUL) { if (a < v) stack.push_back(a); if (b < v) stack.push_back(b); }
                else if (op == ZC_NEG) { if (a < v) stack.push_back(a); }
            }
            std::vector<uint32_t> users(main_width, 0);
            std::vector<std::vector<uint8_t>> in_prod(np->polys.size(), std::vector<uint8_t>(main_width, 0)), in_any = in_prod;
            for (size_t pi = 0; pi < np->polys.size(); pi++) {
                const ZcPoly& pl = np->polys[pi];
                auto mark = [&](const ZcLinForm& f, bool prod) { for (uint32_t c : f.cols) if (c < main_width) { in_any[pi][c] = 1; if (prod) in_prod[pi][c] = 1; } };
                for (const ZcPolyTerm& t : pl.terms) for (int f = 0; f < 3; f++) for (auto& lf : t.f[f]) mark(lf, true);
                for (auto& f : pl.rest) mark(f, false);
                for (uint32_t c = 0; c < main_width; c++) users[c] += in_any[pi][c];
            }
            for (size_t pi = 0; pi < np->polys.size(); pi++)
                for (uint32_t c = 0; c < main_width; c++)
                    if (in_any[pi][c] && !in_prod[pi][c] && !interp[c] && users[c] == 1) np->polys[pi].owned.push_back(c);
        }
        for (const ZcPoly& pl : np->polys) np->poly_segs.push_back(zc_poly_segments(pl));
        auto drop_hinted = [&](std::vector<uint32_t>& sch) {      // asserts carry their constraint index in operand b by now
            if (np->macros.empty()) return;
            for (size_t k = 0; k < sch.size() / 3; k++)
                if (sch[3 * k] == ZC_ASSERT_ZERO && hinted(sch[3 * k + 2])) { sch[3 * k] = ZC_CONST; sch[3 * k + 1] = 0; sch[3 * k + 2] = 0; }
        };
        // fold constants into immediates, then pick the instruction order with the smallest register file
        std::vector<uint32_t> folded, sched;
        fold_immediates(program, n_instr, &folded);
        static const int forced_mode = [] { const char* e = getenv("SP1HIP_ZC_SCHEDULE"); return e ? atoi(e) : -1; }();
        SP1HIP_REQUIRE(forced_mode <= 3, "SP1HIP_ZC_SCHEDULE must be 0, 1, 2 or 3 (a debug knob; unset = try them)");
        // mode 3 (rematerialised loads: a ~3x longer program with a much smaller file) only where the file is the problem: when the
        // best of the other orders needs at least this many registers (SP1HIP_ZC_LAZY_MIN_REGS; 64 = two waves' files per CU). It was
        // 128 while the secp256k1 / uint256 chips (195 - 225) were the only ones above 40; the tower / carry chips that came later
        // (Bn254FpOpAssign 103, Uint256Ops 71, Bn254Fp2AddSubAssign 66: the same FieldOpCols programs) take the same form at 64;
        // every chip with a measured schedule is below 40 and keeps it
        static const uint32_t lazy_min_regs = [] { const char* e = getenv("SP1HIP_ZC_LAZY_MIN_REGS"); return e ? (uint32_t)strtoul(e, nullptr, 10) : 64u; }();
        uint32_t best_regs = 0xffffffffu;
        for (int mode = 0; mode < 4; mode++) {
            if (forced_mode >= 0 && mode != forced_mode) continue;
            if (forced_mode < 0 && mode == 3 && best_regs < lazy_min_regs) continue;
            std::vector<uint32_t> cand;
            std::vector<Chunk> mono;
            schedule_program(folded.data(), n_instr, main_width, mode, &cand);
            std::vector<uint32_t> cand_f = cand;
            drop_hinted(cand_f);
            SP1HIP_TRY(build_chunks(cand_f.data(), (uint32_t)(cand_f.size() / 3), main_width, prep_width, 0xffffffffu, &mono, ZC_CHUNK_HARD_MAX, &np->macros, &np->polys));
            uint32_t regs = 0;
            for (auto& ck : mono) regs = std::max(regs, ck.n_regs);
            if (regs < best_regs) { best_regs = regs; sched.swap(cand); np->mono.swap(mono); }
        }
        const uint32_t n_sched = (uint32_t)(sched.size() / 3);
        static const bool zc_debug = getenv("SP1HIP_ZC_DEBUG") != nullptr;
        if (zc_debug) {
            size_t mono_instr = 0;
            for (auto& ck : np->mono) mono_instr += ck.prog.size() / 4;
            fprintf(stderr, "[sp1hip zc] chip %d: %u ssa instrs, %u+%u cols -> undivided program %zu words, %u registers\n",
                    chip_index, n_instr, main_width, prep_width, mono_instr, best_regs);
        }
        SP1HIP_TRY(allocate_registers(sched.data(), n_sched, &np->prog, &np->n_regs));
        static const uint32_t chunk_limit = [] { const char* e = getenv("SP1HIP_ZC_CHUNK_LIMIT"); return e ? std::max<uint32_t>((uint32_t)atoi(e), 8u) : ZC_CHUNK_LIMIT; }();
        static const uint32_t chunk_hard = [] { const char* e = getenv("SP1HIP_ZC_CHUNK_HARD_MAX"); return e ? std::max<uint32_t>((uint32_t)atoi(e), 8u) : ZC_CHUNK_HARD_MAX; }();
        std::vector<uint32_t> sched_f = sched;
        drop_hinted(sched_f);
        SP1HIP_TRY(build_chunks(sched_f.data(), n_sched, main_width, prep_width, chunk_limit, &np->chunks, chunk_hard, &np->macros, &np->polys));
        SP1HIP_TRY(build_chunks(sched_f.data(), n_sched, main_width, prep_width, ZC_FINE_LIMIT, &np->fine, ZC_FINE_LIMIT, &np->macros, &np->polys));
        np->sched = sched;
        // trust, but verify: on a pseudo-random row the fused pieces must give what the caller's SSA gives for the constraints
        // they replace (a hint on the wrong columns, or on constraints that are not the Poseidon2 sub-AIR, is an error here)
        if (!np->macros.empty()) {
            std::vector<uint32_t> row(main_width), prow(std::max<uint32_t>(prep_width, 1u)), want(asserts_total(program, n_instr), 0u);
            uint64_t x = 0x9E3779B97F4A7C15ull ^ h;
            auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x % kb::P); };
            for (auto& v : row) v = rnd();
            for (auto& v : prow) v = rnd();
            uint32_t n_pub = 1;
            for (uint32_t k = 0; k < n_instr; k++) if (program[3 * k] == ZC_PUBLIC) n_pub = std::max(n_pub, program[3 * k + 1] + 1);
            std::vector<uint32_t> pub(n_pub, 0u);
            eval_words_row(np->prog.data(), np->prog.size() / 4, np->n_regs, row.data(), prow.data(), pub.data(),
                           [&](uint32_t idx, uint32_t v) { if (idx < want.size()) want[idx] = v; });
            for (const ZcMacro& m : np->macros) {
                bool ok = true;
                uint32_t n_seen = 0;
                macro_eval_row(m, np->polys, row.data(), [&](uint32_t j, uint32_t v) {
                    n_seen++;
                    const bool same = m.first_constraint + j < want.size() && want[m.first_constraint + j] == v;
                    if (!same && zc_debug)
                        fprintf(stderr, "[sp1hip zc] hint kind %u: constraint %u + %u: pieces give %08x, the program %08x\n", m.kind, m.first_constraint, j,
                                v, m.first_constraint + j < want.size() ? want[m.first_constraint + j] : 0u);
                    ok &= same;
                });
                SP1HIP_REQUIRE(ok && n_seen == m.n_constraints(), "a fused-kernel hint does not match the constraints it annotates");
            }
        }
        plan = np;
        std::lock_guard<std::mutex> lk(plan_mutex);
        if (plan_cache.size() > 4096) plan_cache.clear();
        plan_cache[h] = plan;
    }
    *out = plan;
    return SP1HIP_SUCCESS;
}

// register-file bytes one lane needs in LDS
template <bool FIRST> static inline size_t zc_rf_lane_bytes(uint32_t n_regs) { return (size_t)n_regs * (FIRST ? 4 : 16); }
constexpr size_t ZC_LDS_CU = 160 * 1024;          // LDS of a gfx950 compute unit; one workgroup may declare all of it
constexpr size_t ZC_LDS_BUDGET = ZC_LDS_CU;

// The workgroup width (256 / 128 / 64 lanes) that keeps the most lanes resident per compute unit given what the LDS register
// file costs: a CU holds floor(160 KB / LDS per workgroup) workgroups, so a file of R extension registers (16 R bytes per lane)
// allows ~10240 / R lanes however they are grouped, and narrower workgroups pack the remainder better (R = 50: three 64-lane
// workgroups = 192 lanes against one 128-lane workgroup). Ties go to the wider workgroup. 0 if even one wave's file does not
// fit 160 KB (R > 160 in the extension rounds): the caller then runs the chip's finer chunks.
template <bool FIRST> static inline uint32_t zc_wg_for(uint32_t n_regs, size_t other_lds) {
    // Default (end of round 5): the occupancy rule above — the workgroup width that puts the most lanes on a CU. SP1HIP_ZC_WG=legacy:
    // the widest workgroup whose file fits 64 KB (the rule every measurement of rounds 2-4 was taken with), the occupancy rule
    // only for files beyond that. Measured with two fork streams: fibonacci shard 86.4-86.8 either way, recorded-shape shard
    // 74.7 -> 74.2-74.6 (a 15-register program runs 640 lanes per CU instead of 512, a 7-register one 2,048 instead of 1,280)
    const bool occ = [] { const char* e = getenv("SP1HIP_ZC_WG"); return !(e && e[0] == 'l'); }();     // default: occupancy; "legacy": the 64 KB rule
    if (!occ)
        for (uint32_t wg = 256; wg >= 64; wg >>= 1)
            if (other_lds + zc_rf_lane_bytes<FIRST>(n_regs) * wg <= 64 * 1024) return wg;
    uint32_t best = 0, best_lanes = 0;
    for (uint32_t wg = 256; wg >= 64; wg >>= 1) {
        const size_t per_wg = other_lds + zc_rf_lane_bytes<FIRST>(n_regs) * wg;
        if (per_wg > ZC_LDS_CU) continue;
        const uint32_t lanes = (uint32_t)std::min<size_t>(ZC_LDS_CU / per_wg, 2048 / wg) * wg;      // also: 32 waves per CU
        if (lanes > best_lanes) { best_lanes = lanes; best = wg; }
    }
    return best;
}

// One group of descriptors = a contiguous block range [block_lo, block_lo + n_blocks) launched together.
template <bool FIRST>
static int launch_round(uint32_t max_regs, bool staged, bool fused, const ZcDesc* d_descs, int n_descs, uint32_t block_lo, uint32_t n_blocks,
                        uint32_t max_instr, const uint32_t* eq, uint32_t eq_len, const uint32_t* publics, uint32_t* partial, hipStream_t s) {
    const size_t lds = 32 * 4 + (staged ? (size_t)max_instr * 16 : 0);
    dim3 grid(fused ? n_blocks : n_blocks * 3);      // unfused: workgroup 3 b + p = node p of block b
    const uint32_t ff = fused ? 1u : 0u;
    const uint32_t wg = zc_wg_for<FIRST>(max_regs, lds);
    SP1HIP_REQUIRE(wg != 0, "internal: a register file that does not fit LDS reached the launch (plan_round cuts such programs finer)");
    const size_t total = lds + zc_rf_lane_bytes<FIRST>(max_regs) * wg;
    if (staged) {
        auto kern = zc_round_kernel<FIRST, 0, true>;
        if (total > 48 * 1024) SP1HIP_TRY(ensure_dynamic_lds((const void*)kern, (int)ZC_LDS_BUDGET));
        hipLaunchKernelGGL(kern, grid, dim3(wg), total, s, d_descs, n_descs, eq, eq_len, publics, partial, (uint32_t)(lds / 4), block_lo, ff);
    } else {
        auto kern = zc_round_kernel<FIRST, 0, false>;
        if (total > 48 * 1024) SP1HIP_TRY(ensure_dynamic_lds((const void*)kern, (int)ZC_LDS_BUDGET));
        hipLaunchKernelGGL(kern, grid, dim3(wg), total, s, d_descs, n_descs, eq, eq_len, publics, partial, (uint32_t)(lds / 4), block_lo, ff);
    }
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
// basefold.hip: every prefix table of eq over the first t coordinates of a point, t = 0..d, one launch
// the same for the bivariate kernels: three node-group workgroups per block of row quads, the extension rounds' register file
static int launch_biv_round(uint32_t max_regs, bool staged, const ZcDesc* d_descs, int n_descs, uint32_t block_lo, uint32_t n_blocks,
                            uint32_t max_instr, const uint32_t* eq, uint32_t eq_len, const uint32_t* publics, uint32_t* partial, hipStream_t s) {
    const size_t lds = 32 * 4 + (staged ? (size_t)max_instr * 16 : 0);
    const dim3 grid(n_blocks * ZC_BIV_GROUPS);
    const uint32_t wg = zc_wg_for<false>(max_regs, lds);      // 16-byte slots: four node values per register
    SP1HIP_REQUIRE(wg != 0, "internal: a register file that does not fit LDS reached the launch (plan_round cuts such programs finer)");
    const size_t total = lds + zc_rf_lane_bytes<false>(max_regs) * wg;
    if (staged) {
        auto kern = zc_biv_round_kernel<0, true>;
        if (total > 48 * 1024) SP1HIP_TRY(ensure_dynamic_lds((const void*)kern, (int)ZC_LDS_BUDGET));
        hipLaunchKernelGGL(kern, grid, dim3(wg), total, s, d_descs, n_descs, eq, eq_len, publics, partial, (uint32_t)(lds / 4), block_lo);
    } else {
        auto kern = zc_biv_round_kernel<0, false>;
        if (total > 48 * 1024) SP1HIP_TRY(ensure_dynamic_lds((const void*)kern, (int)ZC_LDS_BUDGET));
        hipLaunchKernelGGL(kern, grid, dim3(wg), total, s, d_descs, n_descs, eq, eq_len, publics, partial, (uint32_t)(lds / 4), block_lo);
    }
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int eq_prefix_tables_soa_async(const kb::Ext* h_point, int d, uint32_t* d_out, hipStream_t s);
}

// fork streams a round's launches are spread over, besides the caller's own (SP1HIP_ZC_NFORK = 1..3). Default 2 since the end of
// round 5: with three (+ the caller's = the four hardware queues a process gets) the commit's side stream shares a queue with one of
// them; measured A/B/A/B on one box, whole proof: fibonacci shard 89.2 -> 86.4-86.8 ms, recorded-shape shard 76.8 -> 74.7 (one
// fork: 88.0 / —)
static int zc_fork_streams() {
    static const int n = [] { const char* e = getenv("SP1HIP_ZC_NFORK"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v > 3 ? 3 : v; }();
    return n;
}

static int zerocheck_prove_impl(const sp1hip_zc_chip_t* chips, int n_chips, int max_log_row_count,
                               const sp1hip_ext_t* h_zeta, const sp1hip_ext_t* h_openings, sp1hip_ext_t alpha_c,
                               sp1hip_ext_t gkr_c, const uint32_t* h_publics, int n_publics,
                               sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                               sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(chips && n_chips > 0 && h_zeta && h_openings && challenger && proof_len, "null argument");
    SP1HIP_REQUIRE(max_log_row_count >= 1 && max_log_row_count <= 30, "max_log_row_count out of range");
    SP1HIP_REQUIRE(h_publics || n_publics == 0, "null publics");
    ActiveProver active;                                     // a stand-alone call counts as a prover too
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
    // SP1HIP_ZC_TIMING=1: host wall time of the call's three parts on stderr (set-up before the first round | rounds | proof)
    const bool zc_timing = [] { const char* e = getenv("SP1HIP_ZC_TIMING"); return e && e[0] == '1'; }();
    const auto zc_t0 = std::chrono::steady_clock::now();
    auto zc_t1 = zc_t0, zc_t2 = zc_t0;
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
    RoundSyncHost rsync;                          // its counters carry the reduce kernel's last-workgroup ticket
    SP1HIP_TRY(rsync.init(s));
    PinnedStage stage;                            // small uploads go through a pinned block (round_sync.hpp)
    SP1HIP_TRY(stage.init(s));
    if (n_publics) SP1HIP_TRY(stage.upload(d_publics.p, publics.data(), (size_t)n_publics * 4));
    std::vector<uint32_t> blob;
    Ext rho = kb::ext_zero();                     // 1 / alpha, once a chip needs it
    bool have_rho = false;
    std::vector<kb::Ext> poly_scratch[2];
    std::vector<std::unique_ptr<ChipState>> st;
    std::vector<Ext> claims;
    size_t oo = 0;
    for (int i = 0; i < n_chips; i++) {
        std::unique_ptr<ChipState> c(new ChipState());
        c->in = &chips[i];
        uint32_t asserts = 0;
        for (uint32_t k = 0; k < chips[i].n_instr; k++) {
            const uint32_t op = chips[i].program[3 * k], a = chips[i].program[3 * k + 1];
            SP1HIP_REQUIRE(op <= ZC_ASSERT_ZERO || op == ZC_HINT, "bad opcode in constraint program");
            if (op == ZC_ASSERT_ZERO) asserts++;
            if (op == ZC_LOAD_MAIN) SP1HIP_REQUIRE(a < chips[i].main_width, "main column out of range");
            if (op == ZC_LOAD_PREP) SP1HIP_REQUIRE(a < chips[i].prep_width, "preprocessed column out of range");
            if (op == ZC_PUBLIC) SP1HIP_REQUIRE((int)a < n_publics, "public value index out of range");
        }
        SP1HIP_REQUIRE(asserts == chips[i].num_constraints, "num_constraints does not match the program");
        {
            std::shared_ptr<const ZcPlan> plan;
            SP1HIP_TRY(zc_get_plan(chips[i].program, chips[i].n_instr, chips[i].main_width, chips[i].prep_width, i, &plan, chips[i].real_rows));
            c->prog = plan->prog; c->n_regs = plan->n_regs; c->chunks = plan->chunks; c->mono = plan->mono; c->fine = plan->fine;
            c->macros = plan->macros;
            c->plan = plan;
        }
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
        for (auto& ck : c->mono) {
            c->mono_off.push_back((uint32_t)((blob.size() - c->off_prog) / 4));
            blob.insert(blob.end(), ck.prog.begin(), ck.prog.end());
        }
        for (auto& ck : c->fine) {
            c->fine_off.push_back((uint32_t)((blob.size() - c->off_prog) / 4));
            blob.insert(blob.end(), ck.prog.begin(), ck.prog.end());
        }
        pad4();
        c->off_alpha = blob.size();
        for (const Ext& e : c->alpha_pows) blob.insert(blob.end(), e.c, e.c + 4);
        c->off_gkr = blob.size();
        for (const Ext& e : c->gkr_pows) blob.insert(blob.end(), e.c, e.c + 4);
        // the polynomial identities' affine forms, collapsed for this proof's alpha (zc_poly.hpp)
        c->poly_off.assign(c->macros.size(), 0);
        for (size_t mi = 0; mi < c->macros.size(); mi++) {
            const ZcMacro& m = c->macros[mi];
            if (m.kind != ZC_HINT_POLY) continue;
            if (!have_rho) {
                SP1HIP_REQUIRE(!kb::ext_eq(alpha, kb::ext_zero()), "the batching challenge is zero");
                rho = kb::ext_inv(alpha); have_rho = true;
            }
            while (blob.size() & 7) blob.push_back(0);
            c->poly_off[mi] = blob.size();
            zc_poly_table(c->plan->polys[m.aux0], c->plan->poly_segs[m.aux0], c->alpha_pows.data() + m.first_constraint, rho, &blob, poly_scratch);
        }
        st.push_back(std::move(c));
    }
    DevBuf d_blob;
    SP1HIP_TRY(d_blob.alloc(std::max<size_t>(blob.size(), 4) * 4, s));
    SP1HIP_TRY(stage.upload(d_blob.p, blob.data(), blob.size() * 4));
    for (auto& c : st) {
        c->p_prog = d_blob.u32() + c->off_prog;
        c->p_alpha = d_blob.u32() + c->off_alpha;
        c->p_gkr = d_blob.u32() + c->off_gkr;
        c->p_blob = d_blob.u32();
    }

    std::vector<Ext> zeta(L);
    memcpy(zeta.data(), h_zeta, (size_t)L * 16);
    const Ext lambda = challenger_sample_ext(challenger);
    // eq(zeta[0 .. t), .) for every t < L in ONE launch (basefold.hip: table t is an ext SoA of length 2^t at word offset
    // 4 (2^t - 1)); round r reads table L - r - 1. Three small launches per round before.
    DevBuf d_eq_all;
    SP1HIP_TRY(d_eq_all.alloc(((size_t)1 << L) * 16, s));
    SP1HIP_TRY(eq_prefix_tables_soa_async(zeta.data(), L - 1, d_eq_all.u32(), s));
    std::vector<UniPoly> msgs;
    std::vector<Ext> point;   // [alpha_last, ..., alpha_first]
    std::vector<Ext> round_claims = claims;
    std::vector<std::array<uint32_t, 16>> sums(n_chips);
    std::vector<uint32_t> h_sums((size_t)n_chips * 64);      // up to four reduction ranges per chip (interpreter + one per kind of fused piece)
    DevBuf d_partial, d_sums;
    size_t partial_cap = 0;
    SP1HIP_TRY(d_sums.alloc((size_t)n_chips * 256, s));
    // The folded extension tables of all chips live in two ping-pong buffers sized once (round r writes half r & 1;
    // every round's tables are half the size of the previous round's): no allocation inside the round loop — it used to
    // be ~66 arena calls per round.
    DevBuf d_fold[2];
    {
        size_t words[2] = {4, 4};
        for (int half = 0; half < 2; half++)
            for (int i = 0; i < n_chips; i++) {
                uint64_t rows = chips[i].real_rows;
                for (int k = 0; k < half && rows; k++) rows = (rows + 1) / 2;
                if (rows == 0) continue;
                const uint64_t out_rows = (rows + 1) / 2;
                for (uint32_t width : {chips[i].main_width, chips[i].prep_width})
                    if (width) words[half] += (((size_t)out_rows * width * 4 + 3) & ~(size_t)3);
            }
        SP1HIP_TRY(d_fold[0].alloc(words[0] * 4, s));
        SP1HIP_TRY(d_fold[1].alloc(words[1] * 4, s));
    }
    zc_t1 = std::chrono::steady_clock::now();
    auto zc_iter_t = zc_t1;
    double zc_plan_ms = 0, zc_wait_ms = 0, zc_uni_ms = 0;
    // ---- one round's messages from the chips' values at 0, 2, 4 (sum_as_poly.rs:L187-L287), the transcript, the state behind it.
    // hv[i][k]: chip i's round polynomial at X = 0, 2, 4 WITHOUT the eq factor of the variable being bound (`last` = its zeta
    // coordinate); the value at 1 comes from the chip's running claim. sum_as_poly interpolates through {0, 1, 2, 4, b} with the
    // value at b equal to zero. Closed form, no allocation: the Lagrange basis polynomial of node x_k in {0, 1, 2, 4} is
    // C_k(X) (X - b) / (x_k - b), with C_k the basis polynomial of x_k among those four nodes alone — constants of the field:
    //   C_0 = (X^3 - 7 X^2 + 14 X - 8) / -8, C_1 = (X^3 - 6 X^2 + 8 X) / 3, C_2 = (X^3 - 5 X^2 + 4 X) / -4, C_3 = (X^3 - 3 X^2 + 2 X) / 24
    // so a chip's univariate is (X - b) sum_k z_k C_k(X) with z_k = y_k / (x_k - b); the four inverses come from one
    // inversion (Montgomery's trick). Exact field arithmetic: the same polynomial as any other interpolation.
    static const struct CubicBasis {
        uint32_t c[4][4];                                     // c[k][d]: coefficient of X^d in C_k (Montgomery base words)
        CubicBasis() {
            const int num[4][4] = {{-8, 14, -7, 1}, {0, 8, -6, 1}, {0, 4, -5, 1}, {0, 2, -3, 1}};
            const int den[4] = {-8, 3, -4, 24};
            for (int k = 0; k < 4; k++) {
                const uint32_t dm = kb::to_monty(den[k] < 0 ? kb::P - (uint32_t)(-den[k]) : (uint32_t)den[k]);
                const uint32_t dinv = kb::ext_inv(kb::ext_from_base(dm)).c[0];
                for (int d = 0; d < 4; d++) {
                    const uint32_t nm = kb::to_monty(num[k][d] < 0 ? kb::P - (uint32_t)(-num[k][d]) : (uint32_t)num[k][d]);
                    c[k][d] = kb::mul(nm, dinv);
                }
            }
        }
    } cubic;
    // The interpolation is the SAME linear map for every chip (its nodes and b depend on the round only), so the transcript's
    // message — the lambda-combination of the chips' polynomials — is the interpolation of the lambda-combined node values: four
    // products per chip and ONE interpolation stand between the round's sums and the challenge, instead of an interpolation per
    // chip (34 chips: ~35 us of host arithmetic per round with the device idle behind it). A chip's next claim u_i(a_r) is again
    // linear in its node values, sum_k y_k e_k with e_k = (a_r - b) C_k(a_r) / (x_k - b): four more products per chip, taken
    // AFTER the table update of the round has been launched (update_claims).
    std::vector<Ext> lam_pows(n_chips);                      // chip i's weight lambda^(n_chips - 1 - i): rlc = (...(u_0 l + u_1) l + ...) + u_last
    { Ext cur = kb::ext_one(); for (int i = n_chips - 1; i >= 0; i--) { lam_pows[i] = cur; cur = cur * lambda; } }
    std::vector<std::array<Ext, 4>> ys(n_chips);             // a chip's round polynomial at 0, 1, 2, 4 (with the bound variable's eq factor)
    Ext claim_w[4];                                          // e_k of the round whose claims are pending
    Ext pending_a = kb::ext_zero(), pending_last = kb::ext_zero();
    bool claims_pending = false;
    auto update_claims = [&]() {
        if (!claims_pending) return;
        claims_pending = false;
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            // the variable is bound: the virtual geq polynomial and the eq factor of the bound variables follow (fix_last_variable.rs)
            c.vgeq = c.vgeq.fix(pending_a);
            if (c.rows == 0) { round_claims[i] = kb::ext_zero(); continue; }
            round_claims[i] = ys[i][0] * claim_w[0] + ys[i][1] * claim_w[1] + ys[i][2] * claim_w[2] + ys[i][3] * claim_w[3];
            c.eq_adj = c.eq_adj * (pending_a * pending_last + (kb::ext_one() - pending_a) * (kb::ext_one() - pending_last));
        }
    };
    auto round_messages = [&](const Ext& last, const std::vector<std::array<Ext, 3>>& hv) -> Ext {
        update_claims();                                          // (a caller that did not: the claims of the previous round)
        const Ext b_node = (kb::ext_one() - last) * kb::ext_inv(kb::ext_one() - (last + last));
        Ext inv_xb[4];                                            // 1 / (x_k - b), x = 0, 1, 2, 4
        {
            const Ext dx[4] = {kb::ext_zero() - b_node, kb::ext_one() - b_node, ext_c(2) - b_node, ext_c(4) - b_node};
            const Ext p01 = dx[0] * dx[1], p012 = p01 * dx[2], p0123 = p012 * dx[3];
            Ext run = kb::ext_inv(p0123);
            inv_xb[3] = run * p012; run = run * dx[3];
            inv_xb[2] = run * p01; run = run * dx[2];
            inv_xb[1] = run * dx[0];
            inv_xb[0] = run * dx[1];
        }
        const Ext three = ext_c(3), seven = ext_c(7);
        const Ext f0 = kb::ext_one() - last, f2 = last * three - kb::ext_one(), f4 = last * seven - three;
        Ext Y[4] = {kb::ext_zero(), kb::ext_zero(), kb::ext_zero(), kb::ext_zero()};
        for (int i = 0; i < n_chips; i++) {
            if (st[i]->rows == 0) continue;
            const Ext y0 = hv[i][0] * f0;
            ys[i] = {y0, round_claims[i] - y0, hv[i][1] * f2, hv[i][2] * f4};
            for (int k = 0; k < 4; k++) Y[k] = Y[k] + ys[i][k] * lam_pows[i];
        }
        // (X - b) sum_k z_k C_k(X), z_k = Y_k / (x_k - b): coefficient d of the sum is g[d]
        UniPoly rlc(n_chips ? 5 : 1, kb::ext_zero());
        if (n_chips) {
            const Ext z[4] = {Y[0] * inv_xb[0], Y[1] * inv_xb[1], Y[2] * inv_xb[2], Y[3] * inv_xb[3]};
            Ext g[4];
            for (int d = 0; d < 4; d++) {
                Ext acc = kb::ext_mul_base(z[0], cubic.c[0][d]);
                for (int k = 1; k < 4; k++) acc = acc + kb::ext_mul_base(z[k], cubic.c[k][d]);
                g[d] = acc;
            }
            rlc[0] = kb::ext_zero() - b_node * g[0];
            for (int d = 1; d < 4; d++) rlc[d] = g[d - 1] - b_node * g[d];
            rlc[4] = g[3];
        }
        for (auto& cf : rlc)
            for (int k = 0; k < 4; k++) challenger_observe(challenger, cf.c[k]);
        msgs.push_back(rlc);
        const Ext a_r = challenger_sample_ext(challenger);
        point.insert(point.begin(), a_r);
        // e_k = (a_r - b) C_k(a_r) / (x_k - b)
        {
            const Ext a2 = a_r * a_r, a3 = a2 * a_r, ab = a_r - b_node;
            for (int k = 0; k < 4; k++) {
                const Ext ck = kb::ext_from_base(cubic.c[k][0]) + kb::ext_mul_base(a_r, cubic.c[k][1]) + kb::ext_mul_base(a2, cubic.c[k][2]) + kb::ext_mul_base(a3, cubic.c[k][3]);
                claim_w[k] = ab * ck * inv_xb[k];
            }
        }
        pending_a = a_r; pending_last = last; claims_pending = true;
        return a_r;
    };
    // ---- the plan of a round: which chips run in which form, the descriptors of every launch, the reduction ranges and the table
    // update that ends the round. It depends on the tables' heights and addresses only — not on anything the transcript
    // produces — so round r + 1 is planned and its descriptors uploaded while round r's kernels run (the host used to do
    // this between the fix launch and the round's first launch, ~35 us per round with the device idle).
    struct Group { bool staged; uint32_t wg, resident, max_regs, max_instr, block_lo, n_blocks; std::vector<int> chips; };
    struct RoundPlan {
        std::vector<ZcDesc> descs;
        std::vector<ZcChipRange> ranges;
        std::vector<int> desc_chip;
        std::vector<Group> groups;
        uint32_t total_blocks = 0, macro_lo[ZC_MACRO_KINDS + 1] = {}, macro_n[ZC_MACRO_KINDS + 1] = {};
        bool poly_wave = false;                             // the polynomial identities run one wave per row pair this round (zc_poly_wave_kernel)
        std::vector<ZcFixDesc> fds;
        std::vector<uint32_t*> fresh;
        std::vector<std::pair<int, bool>> owner;
        uint32_t fix_blocks = 0;
        size_t off_ranges = 0, off_fds = 0, pack_bytes = 0;
        std::vector<uint8_t> pack;
        std::vector<uint64_t> rows_next;
        std::vector<const uint32_t*> main_next, prep_next;
    };
    // biv: the plan of rounds 0 AND 1 together (bivariate kernels): the units are row quads, the table update folds by both
    // challenges (zc_fix2_kernel) into the buffer round 1 would have written, the reduction ranges carry the quad of the first padded row
    auto plan_round = [&](int r, const std::vector<uint64_t>& vrows, const std::vector<const uint32_t*>& vmain,
                          const std::vector<const uint32_t*>& vprep, RoundPlan& rp, bool biv) -> int {
        const uint64_t unit = biv ? 4 : 2;                  // rows per term
        // descriptors: one per (chip with real rows, chunk); a chip's blocks are contiguous. Chips are grouped by how
        // their programs run this round — (program staged in LDS?, workgroup width the LDS register file allows) — and
        // every group is one launch over its contiguous block range.
        std::vector<ZcDesc>& descs = rp.descs;
        std::vector<ZcChipRange>& ranges = rp.ranges;
        std::vector<int>& desc_chip = rp.desc_chip;
        std::vector<Group>& groups = rp.groups;
        static const bool mono_enabled = [] { const char* e = getenv("SP1HIP_ZC_MONO"); return !(e && e[0] == '0'); }();
        // SP1HIP_ZC_BIV_CORNERS=inline: the corner sums inside the interpreter's first chunk, as in rounds 4-5 (A/B runs; same bytes)
        static const bool corner_kernel = [] { const char* e = getenv("SP1HIP_ZC_BIV_CORNERS"); return !(e && e[0] == 'i'); }();
        std::vector<char> use_mono(n_chips, 0);
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (vrows[i] == 0) continue;
            const uint32_t terms = (uint32_t)((vrows[i] + unit - 1) / unit);
            uint32_t mono_regs = 1;
            for (auto& ck : c.mono) mono_regs = std::max(mono_regs, ck.n_regs);
            const uint32_t mono_wg = (r == 0 && !biv) ? zc_wg_for<true>(mono_regs, 128) : zc_wg_for<false>(mono_regs, 128);
            // the undivided program pays off when chunking recomputes a lot (long dependency chains shared by many
            // constraints); a program of self-contained constraints runs as chunks in every round: same work, small
            // register files, and as many workgroups as there are constraints
            size_t chunk_words = 0, mono_words = 0;
            for (auto& ck : c.chunks) chunk_words += ck.prog.size();
            for (auto& ck : c.mono) mono_words += ck.prog.size();
            // (a shorter undivided program with the same register file is NOT enough: measured on the term-major Poseidon2
            // program — 2,804 words undivided against 3,808 in 28 chunks, 20 registers either way — the undivided form is
            // 36 % slower: 45 KB of instruction words stream through a 16 KB scalar cache, a chunk's 2-5 KB stay in it)
            use_mono[i] = mono_enabled && c.chunks.size() > 1 && terms >= ZC_MONO_MIN_TERMS && mono_wg != 0 && 2 * chunk_words > 3 * mono_words;
            static const bool fine_enabled = [] { const char* e = getenv("SP1HIP_ZC_FINE"); return !(e && e[0] == '0'); }();
            if (!use_mono[i] && fine_enabled && terms <= ZC_FINE_MAX_TERMS && c.fine.size() > c.chunks.size()) use_mono[i] = 2;
            const std::vector<Chunk>& cks = use_mono[i] == 1 ? c.mono : use_mono[i] == 2 ? c.fine : c.chunks;
            uint32_t regs = 1, instr = 1;
            for (auto& ck : cks) { regs = std::max(regs, ck.n_regs); instr = std::max<uint32_t>(instr, (uint32_t)(ck.prog.size() / 4)); }
            // programs stream through the scalar cache (wave-uniform s_load_dwordx4 straight into SGPRs: no LDS read and no
            // v_readfirstlane per instruction word, and the whole LDS budget goes to the register file). Staging programs
            // of up to SP1HIP_ZC_STAGE_MAX instructions in LDS instead was the default until it was measured 3-5 % slower
            // (recursion shard 14.6 vs 13.8 ms of round kernels, core-shaped 10.4 vs 10.1); the VGPR / scratch tier still stages.
            const uint32_t stage_max = [] { const char* e = getenv("SP1HIP_ZC_STAGE_MAX"); return e ? (uint32_t)atoi(e) : 0u; }();   // read per call (tests)
            bool staged = instr <= stage_max;
            uint32_t wg = (r == 0 && !biv) ? zc_wg_for<true>(regs, staged ? 128 + (size_t)instr * 16 : 128) : zc_wg_for<false>(regs, staged ? 128 + (size_t)instr * 16 : 128);
            if (wg == 0 && use_mono[i] != 2) {      // the file does not fit LDS even for one wave: the finest cut of the program
                use_mono[i] = 2;
                regs = 1; instr = 1;
                for (auto& ck : c.fine) { regs = std::max(regs, ck.n_regs); instr = std::max<uint32_t>(instr, (uint32_t)(ck.prog.size() / 4)); }
                staged = instr <= stage_max;
                wg = (r == 0 && !biv) ? zc_wg_for<true>(regs, staged ? 128 + (size_t)instr * 16 : 128) : zc_wg_for<false>(regs, staged ? 128 + (size_t)instr * 16 : 128);
            }
            SP1HIP_REQUIRE(wg != 0, "constraint program too large: one constraint keeps more than 160 extension values live (the LDS register file of one wave)");
            // A launch allocates its LDS register file for the LARGEST file among its chips, so a group is made of chips that keep
            // the same number of workgroups resident per CU on their own: until round 6 every chip of one workgroup width shared a
            // launch, and SyscallInstrs (32 rows, 23 registers) held the Add / Addi / Sub / Addw chips (7 registers, 4.5 million
            // rows of a fibonacci shard) to 3 workgroups of 128 lanes per CU where their own files allow 11. Chips too short for
            // occupancy to matter (the rounds that run the fine cut) share one group per width whatever their files.
            // SP1HIP_ZC_GROUPS=legacy: one group per width (A/B runs; the proof bytes are the same).
            static const bool by_residency = [] { const char* e = getenv("SP1HIP_ZC_GROUPS"); return !(e && e[0] == 'l'); }();
            uint32_t resident = 0;
            if (by_residency && terms > ZC_FINE_MAX_TERMS) {
                const size_t per_wg = (staged ? 128 + (size_t)instr * 16 : 128) + ((r == 0 && !biv) ? zc_rf_lane_bytes<true>(regs) : zc_rf_lane_bytes<false>(regs)) * wg;
                resident = (uint32_t)std::min<size_t>(ZC_LDS_CU / per_wg, 2048 / wg);
            }
            size_t g = 0;
            for (; g < groups.size(); g++) if (groups[g].staged == staged && groups[g].wg == wg && groups[g].resident == resident) break;
            if (g == groups.size()) groups.push_back(Group{staged, wg, resident, 1, 1, 0, 0, {}});
            groups[g].max_regs = std::max(groups[g].max_regs, regs);
            groups[g].max_instr = std::max(groups[g].max_instr, instr);
            groups[g].chips.push_back(i);
        }
        uint32_t& total_blocks = rp.total_blocks;
        total_blocks = 0;
        for (auto& g : groups) {
            g.block_lo = total_blocks;
            for (int i : g.chips) {
                ChipState& c = *st[i];
                const uint32_t terms = (uint32_t)((vrows[i] + unit - 1) / unit);
                const uint32_t bp = g.wg ? g.wg : 256u;
                uint32_t blocks = (terms + bp - 1) / bp;
                static const uint32_t max_pairs = [] { const char* e = getenv("SP1HIP_ZC_MAX_PAIRS"); return e ? std::max<uint32_t>((uint32_t)atoi(e), 256u) : 131072u; }();
                if (blocks > max_pairs / bp) blocks = std::max(1u, max_pairs / bp);
                const std::vector<Chunk>& cks = use_mono[i] == 1 ? c.mono : use_mono[i] == 2 ? c.fine : c.chunks;
                const std::vector<uint32_t>& offs = use_mono[i] == 1 ? c.mono_off : use_mono[i] == 2 ? c.fine_off : c.chunk_off;
                ZcChipRange rg{total_blocks, 0, biv ? (uint32_t)(vrows[i] / 4) : terms - 1, 0};
                for (size_t q = 0; q < cks.size(); q++) {
                    ZcDesc d{};
                    d.prog = c.p_prog + (size_t)offs[q] * 4;
                    d.n_instr = (uint32_t)(cks[q].prog.size() / 4);
                    d.main = vmain[i]; d.prep = vprep[i]; d.main_w = c.in->main_width; d.prep_w = c.in->prep_width;
                    d.rows = (uint32_t)vrows[i]; d.alpha_pows = c.p_alpha; d.gkr_pows = c.p_gkr;
                    d.block_start = total_blocks; d.n_blocks = blocks;
                    d.alpha_off = cks[q].alpha_off; d.flags = (q == 0 && !(biv && corner_kernel)) ? 1u : 0u;
                    d.block_pairs = bp;
                    total_blocks += blocks;
                    descs.push_back(d);
                }
                rg.n_blocks = total_blocks - rg.block_start;
                ranges.push_back(rg);
                desc_chip.push_back(i);
            }
            g.n_blocks = total_blocks - g.block_lo;
        }
        // the fused pieces of hinted sub-AIRs: one launch per kind (each kind is its own kernel with its own register budget),
        // one block range — and one reduction range — per (kind, chip)
        uint32_t (&macro_lo)[ZC_MACRO_KINDS + 1] = rp.macro_lo;
        uint32_t (&macro_n)[ZC_MACRO_KINDS + 1] = rp.macro_n;
        // the polynomial identities: one lane per row pair while the round is large, one wave per pair once the tallest chip that has
        // them is down to ZC_POLY_WAVE_MAX_TERMS pairs (SP1HIP_ZC_POLY_WAVE=0: always the former; same bytes)
        {
            static const bool wave_on = [] { const char* e = getenv("SP1HIP_ZC_POLY_WAVE"); return !(e && e[0] == '0'); }();
            uint64_t max_terms = 0;
            for (int i = 0; i < n_chips; i++)
                for (const ZcMacro& m : st[i]->macros) if (m.kind == ZC_HINT_POLY && vrows[i]) max_terms = std::max<uint64_t>(max_terms, (vrows[i] + unit - 1) / unit);
            rp.poly_wave = wave_on && !biv && r > 0 && max_terms > 0 && max_terms <= ZC_POLY_WAVE_MAX_TERMS;
        }
        for (uint32_t kind = ZC_HINT_POSEIDON2; kind < ZC_MACRO_KINDS; kind++) {
            if (kind == ZC_MACRO_BOTH_SEPTIC) continue;                   // (a launch shape, not a hint kind)
            macro_lo[kind] = total_blocks;
            for (int i = 0; i < n_chips; i++) {
                ChipState& c = *st[i];
                if (vrows[i] == 0) continue;
                const uint32_t terms = (uint32_t)((vrows[i] + unit - 1) / unit);
                const bool wave_form = kind == ZC_HINT_POLY && rp.poly_wave;
                const uint32_t blocks = wave_form ? std::min<uint32_t>((terms + 3) / 4, 1024u) : std::min<uint32_t>((terms + 255) / 256, 512u);
                ZcChipRange rg{total_blocks, 0, biv ? (uint32_t)(vrows[i] / 4) : terms - 1, 0};
                for (size_t mi = 0; mi < c.macros.size(); mi++) {
                    const ZcMacro& m = c.macros[mi];
                    if (m.kind != kind) continue;
                    for (uint32_t q = 0; q < m.n_pieces(); q++) {
                        ZcDesc d{};
                        if (kind == ZC_HINT_POLY) d.prog = c.p_blob + c.poly_off[mi];
                        d.main = vmain[i]; d.prep = vprep[i]; d.main_w = c.in->main_width; d.prep_w = c.in->prep_width;
                        d.rows = (uint32_t)vrows[i]; d.alpha_pows = c.p_alpha; d.gkr_pows = c.p_gkr;
                        d.block_start = total_blocks; d.n_blocks = blocks;
                        d.alpha_off = m.first_constraint; d.flags = ZC_DESC_MACRO | (q << 8) | (m.kind << 12);
                        d.block_pairs = wave_form ? 4 : 256; d.pad = m.base_col; d.aux0 = m.aux0; d.aux1 = m.aux1;
                            total_blocks += blocks;
                        descs.push_back(d);
                    }
                }
                rg.n_blocks = total_blocks - rg.block_start;
                if (rg.n_blocks) { ranges.push_back(rg); desc_chip.push_back(i); }
            }
            macro_n[kind] = total_blocks - macro_lo[kind];
        }
        // the bivariate rounds' GKR corner sums (zc_biv_corner_kernel): per chip, slices of ZC_CORNER_COLS columns x blocks of 256 quads,
        // one reduction range per chip
        macro_lo[ZC_RANGE_CORNERS] = total_blocks;
        if (biv && corner_kernel)
            for (int i = 0; i < n_chips; i++) {
                ChipState& c = *st[i];
                if (vrows[i] == 0) continue;
                const uint32_t quads = (uint32_t)((vrows[i] + 3) / 4), width = c.in->main_width + c.in->prep_width;
                const uint32_t blocks = std::min<uint32_t>((quads + 255) / 256, 512u);
                ZcChipRange rg{total_blocks, 0, (uint32_t)(vrows[i] / 4), 0};
                for (uint32_t c0 = 0; c0 < width; c0 += ZC_CORNER_COLS) {
                    ZcDesc d{};
                    d.main = vmain[i]; d.prep = vprep[i]; d.main_w = c.in->main_width; d.prep_w = c.in->prep_width;
                    d.rows = (uint32_t)vrows[i]; d.alpha_pows = c.p_alpha; d.gkr_pows = c.p_gkr;
                    d.block_start = total_blocks; d.n_blocks = blocks;
                    d.flags = ZC_DESC_MACRO | (ZC_RANGE_CORNERS << 12);
                    d.block_pairs = 256; d.aux0 = c0; d.aux1 = std::min(width, c0 + ZC_CORNER_COLS);
                    total_blocks += blocks;
                    descs.push_back(d);
                }
                rg.n_blocks = total_blocks - rg.block_start;
                if (rg.n_blocks) { ranges.push_back(rg); desc_chip.push_back(i); }
            }
        macro_n[ZC_RANGE_CORNERS] = total_blocks - macro_lo[ZC_RANGE_CORNERS];
        // the table update that ends this round needs nothing from the transcript but alpha (a kernel argument): plan
        // it now, so that every descriptor of the round goes up in ONE copy
        std::vector<ZcFixDesc>& fds = rp.fds;
        std::vector<uint32_t*>& fresh = rp.fresh;              // the folded tables: slices of the round's half of the ping-pong buffer
        std::vector<std::pair<int, bool>>& owner = rp.owner;   // (chip, is_main)
        uint32_t& fix_blocks = rp.fix_blocks;
        fix_blocks = 0;
        size_t fold_words = 0;
        uint32_t* const fold_base = (uint32_t*)d_fold[biv ? 1 : (r & 1)].p;
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (vrows[i] == 0) continue;
            const uint64_t out_rows = (vrows[i] + unit - 1) / unit;
            for (int which = 0; which < 2; which++) {
                const uint32_t width = which == 0 ? c.in->main_width : c.in->prep_width;
                if (width == 0) continue;
                uint32_t* const nb = fold_base + fold_words;
                fold_words += ((size_t)out_rows * width * 4 + 3) & ~(size_t)3;
                ZcFixDesc fd{};
                fd.in = which == 0 ? vmain[i] : vprep[i];
                fd.out = nb;
                fd.rows = (uint32_t)vrows[i]; fd.width = width; fd.block_start = fix_blocks;
                fd.bpc = (uint32_t)((out_rows + ZC_FIX_ROWS - 1) / ZC_FIX_ROWS);
                fd.bpc_magic = (uint32_t)((((uint64_t)1 << 32) / fd.bpc) & 0xffffffffull);      // (bpc == 1 is special-cased in the kernel)
                fd.n_blocks = fd.bpc * width;
                fix_blocks += fd.n_blocks;
                fds.push_back(fd);
                fresh.push_back(nb);
                owner.push_back({i, which == 0});
            }
        }
        const size_t off_ranges = rp.off_ranges = (descs.size() * sizeof(ZcDesc) + 15) & ~(size_t)15;
        const size_t off_fds = rp.off_fds = (off_ranges + ranges.size() * sizeof(ZcChipRange) + 15) & ~(size_t)15;
        const size_t pack_bytes = rp.pack_bytes = off_fds + fds.size() * sizeof(ZcFixDesc);
        std::vector<uint8_t>& pack = rp.pack;
        pack.assign(std::max<size_t>(pack_bytes, 16), 0);
        if (!descs.empty()) memcpy(pack.data(), descs.data(), descs.size() * sizeof(ZcDesc));
        if (!ranges.empty()) memcpy(pack.data() + off_ranges, ranges.data(), ranges.size() * sizeof(ZcChipRange));
        if (!fds.empty()) memcpy(pack.data() + off_fds, fds.data(), fds.size() * sizeof(ZcFixDesc));
        // the tables the round AFTER this one reads (once the fix launch planned above has run)
        rp.rows_next = vrows; rp.main_next = vmain; rp.prep_next = vprep;
        for (size_t k = 0; k < fds.size(); k++) {
            if (owner[k].second) rp.main_next[owner[k].first] = fresh[k]; else rp.prep_next[owner[k].first] = fresh[k];
        }
        for (int i = 0; i < n_chips; i++) if (rp.rows_next[i]) rp.rows_next[i] = (rp.rows_next[i] + unit - 1) / unit;
        return SP1HIP_SUCCESS;
    };
    // the descriptors of rounds r and r + 1 live in two buffers: round r + 1's go up while round r's launches read theirs
    DevBuf d_descs2[2];
    size_t descs_cap2[2] = {0, 0};
    auto upload_plan = [&](const RoundPlan& rp, int which) -> int {
        if (rp.pack.size() > descs_cap2[which]) {
            d_descs2[which].release();
            descs_cap2[which] = rp.pack.size();
            SP1HIP_TRY(d_descs2[which].alloc(descs_cap2[which], s));
        }
        return stage.upload(d_descs2[which].p, rp.pack.data(), rp.pack_bytes);
    };
    // SP1HIP_ZC_BIVARIATE=0: the sequential first two rounds; SP1HIP_ZC_FORK=0: every launch of a round on the caller's stream
    // (A/B runs and the tests of those paths — read per call; the proof bytes are the same)
    const bool biv_enabled = [] { const char* e = getenv("SP1HIP_ZC_BIVARIATE"); return !(e && e[0] == '0'); }();
    const bool fork_enabled = [] { const char* e = getenv("SP1HIP_ZC_FORK"); return !(e && e[0] == '0'); }();
    const bool biv = biv_enabled && L >= 2;
    std::vector<std::unique_ptr<RoundPlan>> plans;           // (kept until the call returns)
    {
        std::vector<uint64_t> rows0(n_chips);
        std::vector<const uint32_t*> main0(n_chips), prep0(n_chips);
        for (int i = 0; i < n_chips; i++) { rows0[i] = st[i]->rows; main0[i] = st[i]->d_main; prep0[i] = st[i]->d_prep; }
        plans.emplace_back(new RoundPlan());
        if (L > 0) {
            SP1HIP_TRY(plan_round(0, rows0, main0, prep0, *plans.back(), biv));
            SP1HIP_TRY(upload_plan(*plans.back(), biv ? 1 : 0));      // (round 2's plan takes buffer 0 while the bivariate launches still read theirs)
        }
    }
    int r_first = 0;
    if (biv) {
        // ================= rounds 0 and 1 from ONE pass over the base-field traces (see zc_biv_node) =================
        RoundPlan& rp = *plans[0];
        const int nv = L;
        const uint32_t eq_len = 1u << (nv - 2);
        const uint32_t* d_eq = d_eq_all.u32() + 4 * (((size_t)1 << (nv - 2)) - 1);    // eq over the nv - 2 variables of the quad index
        const int n_descs = (int)rp.descs.size(), n_ranges = (int)rp.ranges.size();
        DevBuf& d_descs = d_descs2[1];
        const ZcDesc* dd = (const ZcDesc*)d_descs.p;
        const ZcChipRange* d_ranges_p = (const ZcChipRange*)((const uint8_t*)d_descs.p + rp.off_ranges);
        const ZcFixDesc* d_fix_p = (const ZcFixDesc*)((const uint8_t*)d_descs.p + rp.off_fds);
        std::vector<uint32_t> bsums((size_t)std::max(n_ranges, 1) * ZC_BIV_SUM_WORDS);
        if (n_descs) {
            partial_cap = (size_t)rp.total_blocks * ZC_BIV_NODES * 8 * 4;
            SP1HIP_TRY(d_partial.alloc(partial_cap, s));
            DevBuf d_bsums;
            SP1HIP_TRY(d_bsums.alloc(bsums.size() * 4, s));
            const DeviceCtx* dctx;
            SP1HIP_TRY(get_device_ctx(&dctx));
            {
                ScopedTimer tm("zerocheck_round", s);
                // the launches on fork streams, joined in front of the reduction (as in the later rounds)
                const int n_launches = (int)rp.groups.size() + (rp.macro_n[1] ? 1 : 0) + (rp.macro_n[2] ? 1 : 0) + (rp.macro_n[3] ? 1 : 0) + (rp.macro_n[5] ? 1 : 0) + (rp.macro_n[7] ? 1 : 0);
                const bool forked = fork_enabled && n_launches > 1 && active_provers() <= 1;
                constexpr int N_FORK = 3;
                const int n_fork = zc_fork_streams();
                hipStream_t* fork_s = nullptr;
                hipEvent_t* fork_ev = nullptr;
                if (forked) {
                    SP1HIP_TRY(fork_streams_for(s, N_FORK, &fork_s, &fork_ev));
                    SP1HIP_HIP(hipEventRecord(fork_ev[0], s));
                }
                bool fork_used[N_FORK] = {false, false, false};
                auto stream_of = [&](int slot) -> hipStream_t {         // slot 0: the caller's stream
                    if (!forked || slot == 0) return s;
                    if (slot > n_fork) slot = 1 + (slot - 1) % n_fork;
                    if (!fork_used[slot - 1]) { fork_used[slot - 1] = true; (void)hipStreamWaitEvent(fork_s[slot - 1], fork_ev[0], 0); }
                    return fork_s[slot - 1];
                };
                // these launches fill the device together (the round is throughput-bound: 5.1 ms on the core shard however they are
                // placed): the largest interpreter group on the caller's stream, the Poseidon2 pieces on the second, the other
                // interpreter groups on the third, the septic pieces on the fourth
                // (the interpreter groups, longest first — workgroups x instructions —, each on the less loaded of the caller's stream and
                // the third: since the groups are cut by residency there are up to ten of them, and one stream for all but the largest
                // serialised 6 ms of launches)
                {
                    std::vector<size_t> by_work(rp.groups.size());
                    for (size_t g = 0; g < by_work.size(); g++) by_work[g] = g;
                    auto work = [&](size_t g) { return (double)rp.groups[g].n_blocks * (double)rp.groups[g].max_instr; };
                    std::stable_sort(by_work.begin(), by_work.end(), [&](size_t a, size_t b) { return work(a) > work(b); });
                    double load0 = 0, load2 = 0;
                    for (size_t gi : by_work) {
                        const auto& g = rp.groups[gi];
                        const int slot = load0 <= load2 ? 0 : 2;
                        (slot == 0 ? load0 : load2) += work(gi);
                        SP1HIP_TRY(launch_biv_round(g.max_regs, g.staged, dd, n_descs, g.block_lo, g.n_blocks, g.max_instr, d_eq, eq_len, d_publics.u32(), d_partial.u32(), stream_of(slot)));
                    }
                }
#define SP1HIP_ZC_BIV_MACRO_LAUNCH(KIND, SLOT)                                                                                           \
                if (rp.macro_n[KIND]) {                                                                                                \
                    hipLaunchKernelGGL((zc_biv_macro_kernel<KIND>), dim3(rp.macro_n[KIND] * (uint32_t)ZC_BIV_NODES), dim3(256), 0, stream_of(SLOT), dd, n_descs, d_eq, eq_len, d_partial.u32(), rp.macro_lo[KIND], dctx->d_rc); \
                    SP1HIP_LAUNCH_CHECK();                                                                                             \
                }
                // the second stream: the long fused launches, longest first (a Keccak shard's pieces 3.0 ms, the MulOperation pieces of a
                // fibonacci shard 2.7 ms, Poseidon2 1.0-6.0 ms where the Global chip is tall), then the short ones (the polynomial
                // identities, the GKR corner sums, the septic curve pieces); the third stream: behind its interpreter groups the septic
                // sum pieces (6.6 ms on a shard with 1.7 million Global rows). On such a shard the short launches behind the septic sum
                // were a 1.4 ms tail with the device nearly idle, and in front of it they delay the longest launch by as much.
                if (rp.macro_n[ZC_HINT_KECCAK]) {          // four nodes per pass: three node-group workgroups per block
                    hipLaunchKernelGGL(zc_biv_keccak_kernel, dim3(rp.macro_n[ZC_HINT_KECCAK] * ZC_BIV_GROUPS), dim3(256), 0, stream_of(1), dd, n_descs, d_eq, eq_len, d_partial.u32(), rp.macro_lo[ZC_HINT_KECCAK]);
                    SP1HIP_LAUNCH_CHECK();
                }
                SP1HIP_ZC_BIV_MACRO_LAUNCH(3u, 2)
                SP1HIP_ZC_BIV_MACRO_LAUNCH(6u, 1)
                SP1HIP_ZC_BIV_MACRO_LAUNCH(1u, 1)
                if (rp.macro_n[ZC_HINT_POLY]) {            // all twelve nodes per workgroup
                    hipLaunchKernelGGL(zc_biv_poly_kernel, dim3(rp.macro_n[ZC_HINT_POLY]), dim3(256), 0, stream_of(1), dd, n_descs, d_eq, eq_len, d_partial.u32(), rp.macro_lo[ZC_HINT_POLY]);
                    SP1HIP_LAUNCH_CHECK();
                }
                if (rp.macro_n[ZC_RANGE_CORNERS]) {         // the GKR corner sums: columns in slices
                    hipLaunchKernelGGL(zc_biv_corner_kernel, dim3(rp.macro_n[ZC_RANGE_CORNERS]), dim3(256), 0, stream_of(1), dd, n_descs, d_eq, eq_len, d_partial.u32(), rp.macro_lo[ZC_RANGE_CORNERS]);
                    SP1HIP_LAUNCH_CHECK();
                }
                SP1HIP_ZC_BIV_MACRO_LAUNCH(2u, 1)
#undef SP1HIP_ZC_BIV_MACRO_LAUNCH
                if (forked)
                    for (int k = 0; k < N_FORK; k++)
                        if (fork_used[k]) {
                            SP1HIP_HIP(hipEventRecord(fork_ev[1 + k], fork_s[k]));
                            SP1HIP_HIP(hipStreamWaitEvent(s, fork_ev[1 + k], 0));
                        }
                const bool direct = (size_t)n_ranges * ZC_BIV_SUM_WORDS + 1 <= MAILBOX_WORDS;
                const RoundSync rs_pub = direct ? RoundSync{rsync.d_counter, (volatile uint32_t*)mb.h_slot} : RoundSync{};
                if (direct) { rsync.pending = true; mb.pending = true; }
                hipLaunchKernelGGL(zc_biv_reduce_kernel, dim3(n_ranges * ZC_BIV_NODES), dim3(256), 0, s, d_ranges_p, d_partial.u32(), d_eq, eq_len, d_bsums.u32(), rs_pub, mb.seq + 1);
                SP1HIP_LAUNCH_CHECK();
                // round 2's plan and descriptors behind the running launches
                plans.emplace_back(new RoundPlan());                           // (index 1: round 1 has no plan of its own)
                if (L > 2) {
                    plans.emplace_back(new RoundPlan());
                    SP1HIP_TRY(plan_round(2, rp.rows_next, rp.main_next, rp.prep_next, *plans.back(), false));
                    SP1HIP_TRY(upload_plan(*plans.back(), 0));
                }
                const auto zc_w0 = std::chrono::steady_clock::now();
                if (zc_timing) zc_plan_ms += std::chrono::duration<double, std::milli>(zc_w0 - zc_iter_t).count();
                if (direct) { SP1HIP_TRY(mb.wait_next(bsums.data(), (size_t)n_ranges * ZC_BIV_SUM_WORDS)); rsync.pending = false; }
                else SP1HIP_TRY(mb.fetch(d_bsums.p, (size_t)n_ranges * ZC_BIV_SUM_WORDS, bsums.data()));
                if (zc_timing) { zc_iter_t = std::chrono::steady_clock::now(); zc_wait_ms += std::chrono::duration<double, std::milli>(zc_iter_t - zc_w0).count(); }
            }
        } else {
            plans.emplace_back(new RoundPlan());
            if (L > 2) {
                plans.emplace_back(new RoundPlan());
                SP1HIP_TRY(plan_round(2, rp.rows_next, rp.main_next, rp.prep_next, *plans.back(), false));
                SP1HIP_TRY(upload_plan(*plans.back(), 0));
            }
        }
        // ---- per chip: H(X, Y) on {0, 1, 2, 4}^2. Boolean corners: the GKR term's corner sums (constraints vanish on real rows, a
        // padded row's constant cancels against geq). Elsewhere: A + (bilinear extension of the corner sums) - pad_adj eq[qb] geq_b(X, Y),
        // qb = the quad of the first padded row, geq_b = the bilinear extension of [row >= rows] over that quad (zero when it
        // holds no real row: such a quad is not summed and cancels by itself).
        static const int NODE_X[12] = {0, 0, 1, 1, 2, 2, 2, 2, 4, 4, 4, 4}, NODE_Y[12] = {2, 4, 2, 4, 0, 1, 2, 4, 0, 1, 2, 4};
        auto xi = [](int v) { return v == 4 ? 3 : v; };             // index of a coordinate in {0, 1, 2, 4}
        auto small = [](int64_t v) -> Ext { return ext_c((uint32_t)(((v % (int64_t)kb::P) + (int64_t)kb::P) % (int64_t)kb::P)); };
        std::vector<std::array<std::array<Ext, 4>, 4>> H(n_chips);  // H[i][x index][y index]
        {
            std::vector<std::array<Ext, 17>> cs(n_chips);           // merged sums of a chip's ranges: A_0..11, B_0..3, eq[qb]
            std::vector<char> have(n_chips, 0);
            for (size_t k = 0; k < rp.desc_chip.size(); k++) {
                const int ci = rp.desc_chip[k];
                const uint32_t* src = bsums.data() + k * ZC_BIV_SUM_WORDS;
                for (int e = 0; e < 17; e++) {
                    const Ext v{{src[4 * e], src[4 * e + 1], src[4 * e + 2], src[4 * e + 3]}};
                    if (!have[ci] || e == 16) cs[ci][e] = v; else cs[ci][e] = cs[ci][e] + v;
                }
                have[ci] = 1;
            }
            for (int i = 0; i < n_chips; i++) {
                ChipState& c = *st[i];
                if (c.rows == 0) continue;
                const Ext* A = cs[i].data();
                const Ext G00 = cs[i][12], G01 = cs[i][13], G10 = cs[i][14], G11 = cs[i][15];   // B_e: corner (X, Y) = (e >> 1, e & 1)
                const Ext gx = G10 - G00, gy = G01 - G00, gxy = (G11 - G10) - gy;
                const int m = (int)(c.rows % 4);                    // rows of the boundary quad that are real
                const Ext pe = m ? c.pad_adj * cs[i][16] : kb::ext_zero();
                H[i][0][0] = G00; H[i][0][1] = G01; H[i][1][0] = G10; H[i][1][1] = G11;
                for (int e = 0; e < 12; e++) {
                    const int x = NODE_X[e], y = NODE_Y[e];
                    Ext h = A[e] + G00 + gx * small(x) + gy * small(y) + gxy * small(x * y);
                    if (m) {
                        const int i01 = 1 >= m, i10 = 2 >= m;           // [row 4 qb + k >= rows] for k = 1, 2 (k = 0: real, k = 3: padded)
                        const int64_t gq = (int64_t)x * i10 + (int64_t)y * i01 + (int64_t)x * y * (1 - i10 - i01);
                        h = h - pe * small(gq);
                    }
                    H[i][xi(x)][xi(y)] = h;
                }
            }
        }
        std::vector<std::array<Ext, 3>> hv(n_chips);
        // ---- round 0 binds Y (the last variable): h(t) = (1 - z_X) H(0, t) + z_X H(1, t)
        const Ext zX = zeta[nv - 2], zY = zeta[nv - 1];
        for (int i = 0; i < n_chips; i++) {
            if (st[i]->rows == 0) continue;
            for (int k = 0; k < 3; k++) { const int t = k == 0 ? 0 : k + 1; hv[i][k] = (kb::ext_one() - zX) * H[i][0][t] + zX * H[i][1][t]; }
        }
        const Ext a0 = round_messages(zY, hv);
        update_claims();                                             // (round 1's node values need eq_adj and the claims of round 0)
        // ---- round 1 binds X: h(t) = eq(z_Y, a0) x the cubic through H(t, 0), H(t, 1), H(t, 2), H(t, 4) at a0
        Ext Lk[4];                                                   // C_k(a0)
        for (int k = 0; k < 4; k++) {
            Ext acc = kb::ext_from_base(cubic.c[k][3]);
            for (int d = 2; d >= 0; d--) acc = acc * a0 + kb::ext_from_base(cubic.c[k][d]);
            Lk[k] = acc;
        }
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (c.rows == 0) continue;
            for (int k = 0; k < 3; k++) {
                const int t = k == 0 ? 0 : k + 1;                   // index of 0, 2, 4 in {0, 1, 2, 4}
                const Ext v = H[i][t][0] * Lk[0] + H[i][t][1] * Lk[1] + H[i][t][2] * Lk[2] + H[i][t][3] * Lk[3];
                hv[i][k] = c.eq_adj * v;                            // eq_adj = eq(z_Y, a0) since round_messages
            }
        }
        const Ext a1 = round_messages(zX, hv);
        if (zc_timing) { const auto now = std::chrono::steady_clock::now(); zc_uni_ms += std::chrono::duration<double, std::milli>(now - zc_iter_t).count(); zc_iter_t = now; }
        // ---- the tables folded by both challenges
        if (!rp.fds.empty()) {
            ScopedTimer tm("zerocheck_fix", s);
            hipLaunchKernelGGL(zc_fix2_kernel, dim3(rp.fix_blocks), dim3(256), 0, s, d_fix_p, (int)rp.fds.size(), a0, a1);
            SP1HIP_LAUNCH_CHECK();
            for (size_t k = 0; k < rp.fds.size(); k++) {
                ChipState& c = *st[rp.owner[k].first];
                if (rp.owner[k].second) c.d_main = rp.fresh[k]; else c.d_prep = rp.fresh[k];
            }
        }
        update_claims();                                             // (behind the launch of the table update)
        for (int i = 0; i < n_chips; i++)
            if (st[i]->rows) st[i]->rows = (st[i]->rows + 3) / 4;
        r_first = 2;
    }
    for (int r = r_first; r < L; r++) {
        const int nv = L - r;                       // variables left
        const Ext last = zeta[nv - 1];
        // eq(zeta[0 .. nv-1), .) is shared by every chip with real rows
        struct EqView { uint32_t* p; uint32_t* u32() const { return p; } } d_eq{d_eq_all.u32() + 4 * (((size_t)1 << (nv - 1)) - 1)};
        RoundPlan& rp = *plans[r];
        std::vector<ZcDesc>& descs = rp.descs;
        std::vector<ZcChipRange>& ranges = rp.ranges;
        std::vector<int>& desc_chip = rp.desc_chip;
        std::vector<Group>& groups = rp.groups;
        const uint32_t total_blocks = rp.total_blocks;
        uint32_t (&macro_lo)[ZC_MACRO_KINDS + 1] = rp.macro_lo;
        uint32_t (&macro_n)[ZC_MACRO_KINDS + 1] = rp.macro_n;
        std::vector<ZcFixDesc>& fds = rp.fds;
        std::vector<uint32_t*>& fresh = rp.fresh;
        std::vector<std::pair<int, bool>>& owner = rp.owner;
        const uint32_t fix_blocks = rp.fix_blocks;
        const int n_descs = (int)descs.size(), n_ranges = (int)ranges.size();
        DevBuf& d_descs = d_descs2[r & 1];
        const ZcChipRange* d_ranges_p = (const ZcChipRange*)((const uint8_t*)d_descs.p + rp.off_ranges);
        const ZcFixDesc* d_fix_p = (const ZcFixDesc*)((const uint8_t*)d_descs.p + rp.off_fds);
        (void)ranges;
        if (n_descs) {
            if ((size_t)total_blocks * 24 * 4 > partial_cap) {
                d_partial.release();
                partial_cap = (size_t)total_blocks * 24 * 4;
                SP1HIP_TRY(d_partial.alloc(partial_cap, s));
            }
            ScopedTimer tm("zerocheck_round", s);      // (the reference's SP1_GPU_ZEROCHECK_ROUND_TIMING switch)
            // The launches of a round (one per interpreter group, one per kind of fused piece) read the same tables and write
            // disjoint slots of d_partial: nothing orders them but the stream. They go out on fork streams — a round then
            // costs its LONGEST launch instead of their sum (five launches of 30-70 us each in the last fifteen rounds of a
            // core shard; in the large rounds one launch's tail overlaps the next one's head). SP1HIP_ZC_FORK=0: one stream.
            static const bool fuse_nodes = [] { const char* e = getenv("SP1HIP_ZC_FUSE_NODES"); return e && e[0] == '1'; }();
            const int n_launches = (int)groups.size() + (macro_n[1] ? 1 : 0) + (macro_n[2] ? 1 : 0) + (macro_n[3] ? 1 : 0) + (macro_n[5] ? 1 : 0) + (macro_n[7] ? 1 : 0);
            static const uint32_t fork_max_blocks = [] { const char* e = getenv("SP1HIP_ZC_FORK_MAX_BLOCKS"); return e ? (uint32_t)strtoul(e, nullptr, 10) : ZC_FORK_MAX_BLOCKS; }();
            const bool forked = fork_enabled && n_launches > 1 && total_blocks <= fork_max_blocks && active_provers() <= 1;
            // the round's sums reach the host through the mailbox slot when they fit it (they do for any real machine)
            const bool direct = (size_t)n_ranges * 16 + 1 <= MAILBOX_WORDS;
            const RoundSync rs_pub = direct ? RoundSync{rsync.d_counter, (volatile uint32_t*)mb.h_slot} : RoundSync{};
            constexpr int N_FORK = 3;                  // + the caller's stream = the four hardware queues a process gets by default
            const int n_fork = zc_fork_streams();      // SP1HIP_ZC_NFORK = 1..3 fork streams (A/B knob)
            hipStream_t* fork_s = nullptr;
            hipEvent_t* fork_ev = nullptr;
            bool fork_used[N_FORK] = {false, false, false};
            if (forked) {
                SP1HIP_TRY(fork_streams_for(s, N_FORK, &fork_s, &fork_ev));
                SP1HIP_HIP(hipEventRecord(fork_ev[0], s));           // behind the descriptor upload and the previous round's fix
            }
            // longest expected launch first, each on the stream with the least expected work so far (a fused piece is one
            // long dependent chain per workgroup: ~75 / 50 / 60 us at its floor, an interpreter group ~50; above the floor
            // a launch grows with its workgroups per 1024 resident ones)
            struct Launch { int kind; size_t group; double est; int slot; };           // kind 0: interpreter group, 1..3: fused pieces
            std::vector<Launch> order;
            {
                static const double floor_us[ZC_MACRO_KINDS] = {45.0, 75.0, 45.0, 60.0, 60.0, 120.0, 45.0, 90.0};
                // (an interpreter group's length grows with its longest program too: 100 instructions are the unit the floor was measured at)
                for (size_t g = 0; g < groups.size(); g++) order.push_back({0, g, floor_us[0] * (1.0 + groups[g].n_blocks * 3 / 1024.0 * std::max(groups[g].max_instr, 25u) / 100.0), 0});
                const bool both_septic = forked && r > 0 && macro_n[2] && macro_n[3] && (uint64_t)total_blocks * 3 <= ZC_SMALL_ROUND_WGS;
                for (int kind = 1; kind <= 3; kind++) {
                    if (!macro_n[kind] || (both_septic && kind == 2)) continue;
                    if (both_septic && kind == 3) order.push_back({(int)ZC_MACRO_BOTH_SEPTIC, 0, floor_us[3] * (1.0 + (macro_n[2] + macro_n[3]) * 3 / 1024.0), 0});
                    else order.push_back({kind, 0, floor_us[kind] * (1.0 + macro_n[kind] * 3 / 1024.0), 0});
                }
                if (macro_n[ZC_HINT_KECCAK]) order.push_back({(int)ZC_HINT_KECCAK, 0, floor_us[ZC_HINT_KECCAK] * (1.0 + macro_n[ZC_HINT_KECCAK] * 3 / 1024.0), 0});
                if (macro_n[ZC_HINT_MUL]) order.push_back({(int)ZC_HINT_MUL, 0, floor_us[ZC_HINT_MUL] * (1.0 + macro_n[ZC_HINT_MUL] * 3 / 1024.0), 0});
                if (macro_n[ZC_HINT_POLY]) order.push_back({(int)ZC_HINT_POLY, 0, floor_us[ZC_HINT_POLY] * (1.0 + macro_n[ZC_HINT_POLY] * 3 / 1024.0), 0});
                if (forked) {
                    std::stable_sort(order.begin(), order.end(), [](const Launch& a, const Launch& b) { return a.est > b.est; });
                    double load[N_FORK + 1] = {0, 7, 14, 21};          // (the launches leave the host ~7 us apart)
                    // (in the rounds whose launches fill the device the kernel on the fourth queue starts ~0.84 ms after the others —
                    // profiles/r04_gap_trace_timeline.txt — and four queues still beat three: 2.0 against 2.2 ms in round 2)
                    for (Launch& ln : order) {
                        int best = 0;
                        for (int k = 1; k <= n_fork; k++) if (load[k] < load[best]) best = k;
                        ln.slot = best;
                        load[best] += ln.est;
                    }
                }
            }
            const DeviceCtx* dctx;
            SP1HIP_TRY(get_device_ctx(&dctx));
            const ZcDesc* dd = (const ZcDesc*)d_descs.p;
            const uint32_t eq_len = 1u << (nv - 1);
            for (const Launch& ln : order) {
                hipStream_t ls = s;
                if (forked && ln.slot > 0) {
                    const int k = ln.slot - 1;
                    if (!fork_used[k]) { fork_used[k] = true; SP1HIP_HIP(hipStreamWaitEvent(fork_s[k], fork_ev[0], 0)); }
                    ls = fork_s[k];
                }
                if (ln.kind == 0) {
                    const auto& g = groups[ln.group];
                    // SP1HIP_ZC_FUSE_NODES=1: one workgroup evaluates the three nodes of its rows (rows leave HBM once). Measured
                    // on the core-shaped shard: 12.5 ms of round kernels against 11.0 ms unfused — the re-reads of the unfused
                    // form already meet in the memory-side cache (FETCH_SIZE counts those hits), and fusing costs a third of
                    // the parallelism. Off by default; kept for A/B runs.
                    const bool fused = fuse_nodes && g.n_blocks >= 4096;
                    if (r == 0) SP1HIP_TRY(launch_round<true>(g.max_regs, g.staged, fused, dd, n_descs, g.block_lo, g.n_blocks, g.max_instr, d_eq.u32(), eq_len, d_publics.u32(), d_partial.u32(), ls));
                    else SP1HIP_TRY(launch_round<false>(g.max_regs, g.staged, fused, dd, n_descs, g.block_lo, g.n_blocks, g.max_instr, d_eq.u32(), eq_len, d_publics.u32(), d_partial.u32(), ls));
                    continue;
                }
#define SP1HIP_ZC_MACRO_LAUNCH(KIND)                                                                                                   \
                if (ln.kind == (int)KIND) {                                                                                            \
                    if (r == 0) hipLaunchKernelGGL((zc_macro_kernel<true, KIND>), dim3(macro_n[KIND] * 3), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[KIND], dctx->d_rc); \
                    else hipLaunchKernelGGL((zc_macro_kernel<false, KIND>), dim3(macro_n[KIND] * 3), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[KIND], dctx->d_rc); \
                    SP1HIP_LAUNCH_CHECK();                                                                                             \
                }
                SP1HIP_ZC_MACRO_LAUNCH(1u)
                SP1HIP_ZC_MACRO_LAUNCH(2u)
                SP1HIP_ZC_MACRO_LAUNCH(3u)
                SP1HIP_ZC_MACRO_LAUNCH(6u)
                if (ln.kind == (int)ZC_HINT_POLY) {       // the three nodes per workgroup
                    if (rp.poly_wave) hipLaunchKernelGGL(zc_poly_wave_kernel, dim3(macro_n[ZC_HINT_POLY]), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[ZC_HINT_POLY]);
                    else if (r == 0) hipLaunchKernelGGL(zc_poly_kernel<true>, dim3(macro_n[ZC_HINT_POLY]), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[ZC_HINT_POLY]);
                    else hipLaunchKernelGGL(zc_poly_kernel<false>, dim3(macro_n[ZC_HINT_POLY]), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[ZC_HINT_POLY]);
                    SP1HIP_LAUNCH_CHECK();
                }
                if (ln.kind == (int)ZC_HINT_KECCAK) {
                    const bool keccak3 = [] { const char* e = getenv("SP1HIP_ZC_KECCAK3"); return e && e[0] == '1'; }();   // (read per call: tests run both)
                    if (r == 0) hipLaunchKernelGGL((zc_macro_kernel<true, 5u>), dim3(macro_n[5] * 3), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[5], dctx->d_rc);
                    else if (keccak3) hipLaunchKernelGGL(zc_keccak3_kernel, dim3(macro_n[5]), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[5]);
                    else hipLaunchKernelGGL((zc_macro_kernel<false, 5u>), dim3(macro_n[5] * 3), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[5], dctx->d_rc);
                    SP1HIP_LAUNCH_CHECK();
                }
#undef SP1HIP_ZC_MACRO_LAUNCH
                if (ln.kind == (int)ZC_MACRO_BOTH_SEPTIC) {          // (never round 0: that round is far above the small-round bound)
                    hipLaunchKernelGGL((zc_macro_kernel<false, ZC_MACRO_BOTH_SEPTIC>), dim3((macro_n[2] + macro_n[3]) * 3), dim3(256), 0, ls, dd, n_descs, d_eq.u32(), eq_len, d_partial.u32(), macro_lo[2], dctx->d_rc);
                    SP1HIP_LAUNCH_CHECK();
                }
            }
            if (forked)
                for (int k = 0; k < N_FORK; k++)
                    if (fork_used[k]) {
                        SP1HIP_HIP(hipEventRecord(fork_ev[1 + k], fork_s[k]));
                        SP1HIP_HIP(hipStreamWaitEvent(s, fork_ev[1 + k], 0));   // the reduction (and everything after it) follows every launch
                    }
            // the reduce kernel publishes the round's sums itself (ticket on the round-sync counters, payload in the mailbox slot)
            if (direct) { rsync.pending = true; mb.pending = true; }
            if (r == 0) hipLaunchKernelGGL(zc_reduce_kernel<true>, dim3(n_ranges), dim3(256), 0, s, d_ranges_p, d_partial.u32(), d_eq.u32(), eq_len, d_sums.u32(), rs_pub, mb.seq + 1);
            else hipLaunchKernelGGL(zc_reduce_kernel<false>, dim3(n_ranges), dim3(256), 0, s, d_ranges_p, d_partial.u32(), d_eq.u32(), eq_len, d_sums.u32(), rs_pub, mb.seq + 1);
            SP1HIP_LAUNCH_CHECK();
            // the next round's plan and descriptors, behind this round's launches (see plan_round)
            if (r + 1 < L && (int)plans.size() == r + 1) {
                plans.emplace_back(new RoundPlan());
                SP1HIP_TRY(plan_round(r + 1, rp.rows_next, rp.main_next, rp.prep_next, *plans.back(), false));
                SP1HIP_TRY(upload_plan(*plans.back(), (r + 1) & 1));
            }
            const auto zc_w0 = std::chrono::steady_clock::now();
            if (zc_timing) zc_plan_ms += std::chrono::duration<double, std::milli>(zc_w0 - zc_iter_t).count();
            if (direct) { SP1HIP_TRY(mb.wait_next(h_sums.data(), (size_t)n_ranges * 16)); rsync.pending = false; }
            else SP1HIP_TRY(mb.fetch(d_sums.p, (size_t)n_ranges * 16, h_sums.data()));
            if (zc_timing) { zc_iter_t = std::chrono::steady_clock::now(); zc_wait_ms += std::chrono::duration<double, std::milli>(zc_iter_t - zc_w0).count(); }
        }
        if (r + 1 < L && (int)plans.size() == r + 1) {           // (a round without descriptors: nothing was launched above)
            plans.emplace_back(new RoundPlan());
            SP1HIP_TRY(plan_round(r + 1, rp.rows_next, rp.main_next, rp.prep_next, *plans.back(), false));
            SP1HIP_TRY(upload_plan(*plans.back(), (r + 1) & 1));
        }
        {   // a chip with fused pieces has a second range: its sums ADD to the interpreter's (the eq entry is the same)
            std::vector<char> have(n_chips, 0);
            for (size_t k = 0; k < desc_chip.size(); k++) {
                const int ci = desc_chip[k];
                const uint32_t* src = h_sums.data() + k * 16;
                if (!have[ci]) { memcpy(sums[ci].data(), src, 64); have[ci] = 1; }
                else for (int w = 0; w < 12; w++) sums[ci][w] = kb::add(sums[ci][w], src[w]);
            }
        }
        // ---- univariate messages (sum_as_poly.rs:L187-L287): the values of every chip's round polynomial at 0, 2, 4 without the
        // eq factor of the variable being bound
        std::vector<std::array<Ext, 3>> hv(n_chips);
        {
            const Ext two = ext_c(2), four = ext_c(4);
            for (int i = 0; i < n_chips; i++) {
                ChipState& c = *st[i];
                if (c.rows == 0) continue;
                const size_t th = (size_t)((c.rows + 1) / 2) - 1;
                const Ext eq_th{{sums[i][12], sums[i][13], sums[i][14], sums[i][15]}};
                const Ext msb = c.eq_adj * eq_th;
                const Ext y0s{{sums[i][0], sums[i][1], sums[i][2], sums[i][3]}}, y2s{{sums[i][4], sums[i][5], sums[i][6], sums[i][7]}},
                    y4s{{sums[i][8], sums[i][9], sums[i][10], sums[i][11]}};
                const Ext v0 = c.vgeq.fix(kb::ext_zero()).at(th), v2 = c.vgeq.fix(two).at(th), v4 = c.vgeq.fix(four).at(th);
                const Ext pm = c.pad_adj * msb;
                hv[i] = {y0s * c.eq_adj - pm * v0, y2s * c.eq_adj - pm * v2, y4s * c.eq_adj - pm * v4};
            }
        }
        const Ext a_r = round_messages(last, hv);
        if (zc_timing) { const auto now = std::chrono::steady_clock::now(); zc_uni_ms += std::chrono::duration<double, std::milli>(now - zc_iter_t).count(); zc_iter_t = now; }
        if (!fds.empty()) {
            ScopedTimer tm("zerocheck_fix", s);
            if (r == 0) hipLaunchKernelGGL(zc_fix_kernel<true>, dim3(fix_blocks), dim3(256), 0, s, d_fix_p, (int)fds.size(), a_r);
            else hipLaunchKernelGGL(zc_fix_kernel<false>, dim3(fix_blocks), dim3(256), 0, s, d_fix_p, (int)fds.size(), a_r);
            SP1HIP_LAUNCH_CHECK();
            for (size_t k = 0; k < fds.size(); k++) {     // the arena is stream-ordered: the old table is recycled behind this launch
                ChipState& c = *st[owner[k].first];
                if (owner[k].second) c.d_main = fresh[k]; else c.d_prep = fresh[k];
            }
        }
        update_claims();                                           // (behind the launch of the table update: the device is busy again)
        for (int i = 0; i < n_chips; i++)
            if (st[i]->rows) st[i]->rows = (st[i]->rows + 1) / 2;
    }
    update_claims();
    zc_t2 = std::chrono::steady_clock::now();
    // ---- proof: PartialSumcheckProof + per-chip component evaluations (prep then main)
    ByteOut w;
    w.u64((uint64_t)L);
    for (auto& m : msgs) { w.u64(m.size()); for (auto& cf : m) w.ext(cf); }
    Ext claimed = kb::ext_zero(), final_eval = kb::ext_zero();
    for (auto& cl : claims) claimed = claimed * lambda + cl;
    for (int i = 0; i < n_chips; i++) final_eval = final_eval * lambda + round_claims[i];      // (a chip's polynomial of the last round at its challenge)
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
    if (zc_timing) {
        const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[sp1hip zerocheck] set-up %.3f ms | %d rounds %.3f ms (planning + launches %.3f, waiting for the sums %.3f, univariates + transcript %.3f) | openings + proof %.3f ms\n",
                ms(zc_t0, zc_t1), L, ms(zc_t1, zc_t2), zc_plan_ms, zc_wait_ms, zc_uni_ms, ms(zc_t2, std::chrono::steady_clock::now()));
    }
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

// Host-only: plans `program` exactly as sp1hip_zerocheck_prove does and interprets the chosen form of it on one row.
extern "C" int sp1hip_zerocheck_biv_interp_host(uint32_t r00, uint32_t r01, uint32_t r10, uint32_t r11, uint32_t node, uint32_t* out) {
    SP1HIP_REQUIRE(out && node < (uint32_t)ZC_BIV_NODES, "node out of range");
    SP1HIP_REQUIRE(r00 < kb::P && r01 < kb::P && r10 < kb::P && r11 < kb::P, "word not reduced");
    *out = zc_biv_interp(r00, r01, r10, r11, zc_biv_node(node));
    return SP1HIP_SUCCESS;
}

extern "C" int sp1hip_zerocheck_plan_eval(const uint32_t* program, uint32_t n_instr, uint32_t main_width, uint32_t prep_width,
                                          const uint32_t* main_row, const uint32_t* prep_row, const uint32_t* publics,
                                          uint32_t n_publics, int form, uint32_t* out_values, uint32_t n_constraints,
                                          uint32_t* out_stats) {
    SP1HIP_REQUIRE(program || n_instr == 0, "null program");
    SP1HIP_REQUIRE(out_values || n_constraints == 0, "null output");
    SP1HIP_REQUIRE(form >= 0 && form <= 3, "form: 0 whole program, 1 chunks, 2 undivided, 3 fine");
    uint32_t asserts = 0;
    for (uint32_t k = 0; k < n_instr; k++) {
        const uint32_t op = program[3 * k], a = program[3 * k + 1];
        SP1HIP_REQUIRE(op <= ZC_ASSERT_ZERO || op == ZC_HINT, "bad opcode in constraint program");
        if (op == ZC_ASSERT_ZERO) asserts++;
        if (op == ZC_LOAD_MAIN) SP1HIP_REQUIRE(a < main_width && main_row, "main column out of range");
        if (op == ZC_LOAD_PREP) SP1HIP_REQUIRE(a < prep_width && prep_row, "preprocessed column out of range");
        if (op == ZC_PUBLIC) SP1HIP_REQUIRE(a < n_publics && publics, "public value index out of range");
    }
    SP1HIP_REQUIRE(asserts == n_constraints, "n_constraints does not match the program");
    std::shared_ptr<const ZcPlan> plan;
    SP1HIP_TRY(zc_get_plan(program, n_instr, main_width, prep_width, -1, &plan));
    std::vector<uint32_t> seen(n_constraints, 0);
    auto on_assert = [&](uint32_t idx, uint32_t v) { if (idx < n_constraints) { out_values[idx] = v; seen[idx]++; } };
    uint32_t words = 0, pieces = 0, regs = 0;
    if (form == 0) {
        eval_words_row(plan->prog.data(), plan->prog.size() / 4, plan->n_regs, main_row, prep_row, publics, on_assert);
        words = (uint32_t)(plan->prog.size() / 4); pieces = 1; regs = plan->n_regs;
    } else {
        const std::vector<Chunk>& cks = form == 1 ? plan->chunks : form == 2 ? plan->mono : plan->fine;
        for (const Chunk& ck : cks) {
            // (every ASSERT carries its chip-wide constraint index, whatever piece it ended up in)
            eval_words_row(ck.prog.data(), ck.prog.size() / 4, ck.n_regs, main_row, prep_row, publics, on_assert);
            words += (uint32_t)(ck.prog.size() / 4); pieces++; regs = std::max(regs, ck.n_regs);
        }
    }
    if (form != 0) {                                  // the forms the GPU runs: hinted constraints come from the fused pieces
        for (const ZcMacro& m : plan->macros) {
            macro_eval_row(m, plan->polys, main_row, [&](uint32_t j, uint32_t v) { on_assert(m.first_constraint + j, v); });
            pieces += m.n_pieces();
        }
    }
    for (uint32_t k = 0; k < n_constraints; k++) SP1HIP_REQUIRE(seen[k] == 1, "a constraint was not evaluated exactly once");
    if (out_stats) { out_stats[0] = words; out_stats[1] = pieces; out_stats[2] = regs; }
    return SP1HIP_SUCCESS;
}

extern "C" int sp1hip_zerocheck_poly_check(const uint32_t* program, uint32_t n_instr, uint32_t main_width, uint32_t prep_width,
                                           const uint32_t* main_row, sp1hip_ext_t alpha_c, sp1hip_ext_t* out_collapsed,
                                           sp1hip_ext_t* out_direct, uint32_t* n_identities) {
    SP1HIP_REQUIRE(program && main_row && out_collapsed && out_direct && n_identities, "null argument");
    for (uint32_t k = 0; k < n_instr; k++) SP1HIP_REQUIRE(program[3 * k] <= ZC_ASSERT_ZERO || program[3 * k] == ZC_HINT, "bad opcode in constraint program");
    std::shared_ptr<const ZcPlan> plan;
    SP1HIP_TRY(zc_get_plan(program, n_instr, main_width, prep_width, -1, &plan));
    const Ext alpha{{alpha_c.c[0], alpha_c.c[1], alpha_c.c[2], alpha_c.c[3]}};
    SP1HIP_REQUIRE(!kb::ext_eq(alpha, kb::ext_zero()), "the batching challenge is zero");
    const uint32_t n_c = asserts_total(program, n_instr);
    std::vector<Ext> pows(n_c);                                   // [alpha^(n-1), ..., alpha, 1] as in zerocheck_prove_impl
    { Ext cur = kb::ext_one(); for (uint32_t k = n_c; k-- > 0;) { pows[k] = cur; cur = cur * alpha; } }
    const Ext rho = kb::ext_inv(alpha);
    Ext collapsed = kb::ext_zero(), direct = kb::ext_zero();
    std::vector<kb::Ext> scratch[2];
    uint32_t count = 0;
    for (const ZcMacro& m : plan->macros) {
        if (m.kind != ZC_HINT_POLY) continue;
        count++;
        const ZcPoly& pl = plan->polys[m.aux0];
        std::vector<uint32_t> tb;
        zc_poly_table(pl, plan->poly_segs[m.aux0], pows.data() + m.first_constraint, rho, &tb, scratch);
        // the kernels' evaluation of the table on one row: every segment is an affine form (constant first)
        size_t off = ZC_POLY_HDR;
        auto form = [&](uint32_t n, bool with_const) -> Ext {
            Ext f = kb::ext_zero();
            if (with_const) { f = Ext{{tb[off + 4], tb[off + 5], tb[off + 6], tb[off + 7]}}; off += ZC_POLY_ENTRY; }
            for (uint32_t k = 0; k < n; k++, off += ZC_POLY_ENTRY) f = f + kb::ext_mul_base(Ext{{tb[off + 4], tb[off + 5], tb[off + 6], tb[off + 7]}}, main_row[tb[off]]);
            return f;
        };
        Ext v = kb::ext_zero();
        for (uint32_t t = 0; t < tb[0]; t++) {
            Ext p = form(tb[4 + 3 * t], true) * form(tb[5 + 3 * t], true);
            if (tb[6 + 3 * t] != ZC_POLY_NONE) p = p * form(tb[6 + 3 * t], true);
            v = v + p;
        }
        v = v + form(tb[1], true) + form(tb[2], false);
        collapsed = collapsed + v;
        zc_poly_eval_row(pl, main_row, [&](uint32_t k, uint32_t c) { direct = direct + kb::ext_mul_base(pows[m.first_constraint + k], c); });
    }
    memcpy(out_collapsed->c, collapsed.c, 16);
    memcpy(out_direct->c, direct.c, 16);
    *n_identities = count;
    return SP1HIP_SUCCESS;
}

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
