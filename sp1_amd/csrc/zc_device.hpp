// sp1_amd/csrc/zc_device.hpp — device-side pieces of the zerocheck round shared by the bytecode interpreter
// (zerocheck.hip) and the per-chip COMPILED constraint kernels (zc_codegen.cpp generates their source against this
// header; hipcc at build time or hipRTC at run time compiles it). See zerocheck.hip for the reference citations.
#pragma once
#include "kb31.hpp"

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
    uint32_t block_pairs, pad;   // row pairs per block = the width of the workgroups of this chip's launch group; pad: a fused piece's first column
    uint32_t aux0, aux1;         // fused pieces (zc_poseidon2.hpp): second column base / is_real column
};

// Blocks of one chip (all its chunks are contiguous) for the reduction, plus the eq entry it needs.
struct ZcChipRange {
    uint32_t block_start, n_blocks, th, pad;
};

struct ZcFixDesc {
    const uint32_t* in;
    uint32_t* out;
    uint32_t rows, width, block_start, n_blocks;
    uint32_t bpc, bpc_magic;        // workgroups per column = ceil(out_rows / 256), and floor(2^32 / bpc) for the division by it
};

// Table pointers reach the kernels inside descriptors read from memory, so the compiler only knows them as generic
// pointers and would emit FLAT loads; the casts below restore the address space (global / constant).
typedef const uint32_t __attribute__((address_space(1)))* zc_global_words_t;
typedef const uint32_t __attribute__((address_space(4)))* zc_const_words_t;

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
        return ((zc_global_words_t)tbl)[(size_t)col * rows + r];
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
        zc_global_words_t g = (zc_global_words_t)tbl;
#pragma unroll
        for (int k = 0; k < 4; k++) v.c[k] = g[((size_t)col * 4 + k) * rows + r];
        return v;
    }
};

// wave-uniform table entry (alpha / GKR powers: written before the launch, read-only in the kernel): a scalar load
__device__ __forceinline__ kb::Ext load_ext_aos(const uint32_t* p, uint32_t i) {
    const zc_const_words_t c = (zc_const_words_t)(uintptr_t)p;
    return kb::Ext{{c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]}};
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

__device__ __forceinline__ uint32_t zc_wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, __shfl_xor(v, off));
    return v;
}


// extra forms the generated code uses: an operand that is a compile-time constant (Montgomery word)
template <bool FIRST> struct KC;
template <> struct KC<true> {
    static __device__ __forceinline__ uint32_t addc(uint32_t a, uint32_t c) { return kb::add(a, c); }
    static __device__ __forceinline__ uint32_t subc(uint32_t a, uint32_t c) { return kb::sub(a, c); }
    static __device__ __forceinline__ uint32_t csub(uint32_t c, uint32_t a) { return kb::sub(c, a); }
    static __device__ __forceinline__ uint32_t mulc(uint32_t a, uint32_t c) { return kb::mul(a, c); }
};
template <> struct KC<false> {
    static __device__ __forceinline__ kb::Ext addc(kb::Ext a, uint32_t c) { a.c[0] = kb::add(a.c[0], c); return a; }
    static __device__ __forceinline__ kb::Ext subc(kb::Ext a, uint32_t c) { a.c[0] = kb::sub(a.c[0], c); return a; }
    static __device__ __forceinline__ kb::Ext csub(uint32_t c, const kb::Ext& a) {
        return kb::Ext{{kb::sub(c, a.c[0]), kb::neg(a.c[1]), kb::neg(a.c[2]), kb::neg(a.c[3])}};
    }
    static __device__ __forceinline__ kb::Ext mulc(const kb::Ext& a, uint32_t c) { return kb::ext_mul_base(a, c); }
};

}  // namespace sp1hip
