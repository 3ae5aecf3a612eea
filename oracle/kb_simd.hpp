// oracle/kb_simd.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). AVX-512 forms of the oracle's two hottest loops, so that the
// `cpu_baseline` leg of bench.py is not a scalar strawman: sixteen Poseidon2 permutations at once (one per 32-bit
// lane: the leaf hash of sixteen rows, sixteen Merkle compressions) and the Reed-Solomon butterflies sixteen columns at a time.
// Own code of the oracle (nothing shared with sp1_amd/csrc); every function has the scalar form of kb_hash.hpp / kb_pcs.hpp as
// its definition and is checked against it by the same golden vectors (tests/test_oracle_*.py run whichever path the CPU has).
// Runtime dispatch: `simd_available()`; built with per-function target attributes, the library itself stays x86-64-v3.
#pragma once
#include <immintrin.h>

#include "kb_hash.hpp"

namespace orc {
namespace simd {

#define ORC_AVX512 __attribute__((target("avx512f,avx512dq,avx512bw")))

static inline bool simd_available() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512bw") &&
                           !getenv("ORC_NO_SIMD");
    return ok;
}

// sixteen Montgomery words < p per register
ORC_AVX512 static inline __m512i vadd(__m512i a, __m512i b) {
    const __m512i p = _mm512_set1_epi32((int)KB_P);
    const __m512i s = _mm512_add_epi32(a, b);
    return _mm512_min_epu32(s, _mm512_sub_epi32(s, p));
}
ORC_AVX512 static inline __m512i vsub(__m512i a, __m512i b) {
    const __m512i p = _mm512_set1_epi32((int)KB_P);
    const __m512i d = _mm512_sub_epi32(a, b);
    return _mm512_min_epu32(d, _mm512_add_epi32(d, p));
}
// Montgomery product: even and odd lanes through the 32 x 32 -> 64 multiplier, x - (x MU mod 2^32) p >> 32, + p on borrow
ORC_AVX512 static inline __m512i vmul(__m512i a, __m512i b) {
    const __m512i p = _mm512_set1_epi32((int)KB_P), mu = _mm512_set1_epi32((int)KB_MU);
    const __m512i ae = a, ao = _mm512_srli_epi64(a, 32), be = b, bo = _mm512_srli_epi64(b, 32);
    const __m512i xe = _mm512_mul_epu32(ae, be), xo = _mm512_mul_epu32(ao, bo);
    const __m512i te = _mm512_mul_epu32(xe, mu), to = _mm512_mul_epu32(xo, mu);         // low 32 bits of x * MU matter only
    const __m512i ue = _mm512_mul_epu32(te, p), uo = _mm512_mul_epu32(to, p);
    const __m512i de = _mm512_sub_epi64(xe, ue), dod = _mm512_sub_epi64(xo, uo);
    // high halves: (d >> 32) for the even lanes sits in the odd 32-bit slot of de; odd lanes keep their high half in place
    const __m512i hi = _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(de, 32), dod);
    return _mm512_min_epu32(hi, _mm512_add_epi32(hi, p));        // a borrow leaves hi = true + 2^32 - p ... + p wraps it back below p
}
ORC_AVX512 static inline __m512i vcube(__m512i x) { return vmul(vmul(x, x), x); }

ORC_AVX512 static inline void vm4(__m512i* x) {
    const __m512i t01 = vadd(x[0], x[1]), t23 = vadd(x[2], x[3]);
    const __m512i t0123 = vadd(t01, t23);
    const __m512i t01123 = vadd(t0123, x[1]), t01233 = vadd(t0123, x[3]);
    const __m512i n3 = vadd(t01233, vadd(x[0], x[0])), n1 = vadd(t01123, vadd(x[2], x[2]));
    const __m512i n0 = vadd(t01123, t01), n2 = vadd(t01233, t23);
    x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3;
}
ORC_AVX512 static inline void vexternal_linear(__m512i* s) {
    for (int j = 0; j < 16; j += 4) vm4(s + j);
    __m512i sums[4];
    for (int k = 0; k < 4; k++) sums[k] = vadd(vadd(s[k], s[k + 4]), vadd(s[k + 8], s[k + 12]));
    for (int j = 0; j < 16; j++) s[j] = vadd(s[j], sums[j & 3]);
}
// new_i = (sum + d_i s_i) 2^-32 with d = [-2, 1, 2, 4, ..., 2^13, 2^15]: as field operations — sum 2^-32 is one Montgomery product
// by the word 1, d_i 2^-32 is the plain word d_i (a Montgomery product by it)
ORC_AVX512 static inline void vinternal_linear(__m512i* s) {
    static const int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
    __m512i sum = s[0];
    for (int i = 1; i < 16; i++) sum = vadd(sum, s[i]);
    const __m512i one_word = _mm512_set1_epi32(1);
    const __m512i sr = vmul(sum, one_word);
    s[0] = vadd(sr, vmul(s[0], _mm512_set1_epi32((int)(KB_P - 2))));
    for (int i = 1; i < 16; i++) s[i] = vadd(sr, vmul(s[i], _mm512_set1_epi32(1 << SH[i - 1])));
}
ORC_AVX512 static inline void vpermute(__m512i* s) {
    const P2Constants& c = p2c();
    vexternal_linear(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = vcube(vadd(s[i], _mm512_set1_epi32((int)c.ext[r][i].v)));
        vexternal_linear(s);
    }
    for (int r = 0; r < 20; r++) {
        s[0] = vcube(vadd(s[0], _mm512_set1_epi32((int)c.internal[r].v)));
        vinternal_linear(s);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 16; i++) s[i] = vcube(vadd(s[i], _mm512_set1_epi32((int)c.ext[r][i].v)));
        vexternal_linear(s);
    }
}

// leaf digests of rows i0 .. i0 + 15 over the concatenation of row-major tensors (overwrite-mode sponge, rate 8)
struct RowSrc { const F* data; int width; };
ORC_AVX512 static inline void hash_rows16(const RowSrc* ts, size_t n_ts, size_t i0, Digest* out) {
    __m512i s[16];
    for (auto& x : s) x = _mm512_setzero_si512();
    int fill = 0;
    const __m512i lane = _mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    for (size_t t = 0; t < n_ts; t++) {
        const int w = ts[t].width;
        const __m512i idx = _mm512_mullo_epi32(lane, _mm512_set1_epi32(w));
        const F* base = ts[t].data + i0 * (size_t)w;
        for (int c = 0; c < w; c++) {
            s[fill++] = _mm512_i32gather_epi32(idx, (const void*)(base + c), 4);
            if (fill == P2_RATE) { vpermute(s); fill = 0; }
        }
    }
    if (fill) vpermute(s);
    alignas(64) uint32_t tmp[8][16];
    for (int k = 0; k < 8; k++) _mm512_store_si512((void*)tmp[k], s[k]);
    for (int l = 0; l < 16; l++)
        for (int k = 0; k < 8; k++) out[l].d[k].v = tmp[k][l];
}
// next[i0 .. i0 + 15] = compress(prev[2 i], prev[2 i + 1])
ORC_AVX512 static inline void compress16(const Digest* prev, size_t i0, Digest* next) {
    __m512i s[16];
    const __m512i lane = _mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m512i idx = _mm512_slli_epi32(lane, 4);                  // a pair of digests = 16 words
    const uint32_t* base = (const uint32_t*)(prev + 2 * i0);
    for (int k = 0; k < 16; k++) s[k] = _mm512_i32gather_epi32(idx, (const void*)(base + k), 4);
    vpermute(s);
    alignas(64) uint32_t tmp[8][16];
    for (int k = 0; k < 8; k++) _mm512_store_si512((void*)tmp[k], s[k]);
    for (int l = 0; l < 16; l++)
        for (int k = 0; k < 8; k++) next[i0 + l].d[k].v = tmp[k][l];
}
// one butterfly over w columns: a = a + b, b = (a - b) t
ORC_AVX512 static inline void butterfly_row(F* a, F* b, F t, int w) {
    const __m512i tv = _mm512_set1_epi32((int)t.v);
    int c = 0;
    for (; c + 16 <= w; c += 16) {
        const __m512i x = _mm512_loadu_si512((const void*)(a + c)), y = _mm512_loadu_si512((const void*)(b + c));
        _mm512_storeu_si512((void*)(a + c), vadd(x, y));
        _mm512_storeu_si512((void*)(b + c), vmul(vsub(x, y), tv));
    }
    for (; c < w; c++) { const F x = a[c], y = b[c]; a[c] = x + y; b[c] = (x - y) * t; }
}

}  // namespace simd
}  // namespace orc
