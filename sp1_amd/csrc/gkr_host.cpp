// sp1_amd/csrc/gkr_host.cpp — the interaction-variable rounds of a LogUp-GKR layer on the HOST, vectorised (x86-64
// AVX-512; no device code in this translation unit).
//
// Once a layer's row variables are bound, what is left is 2^niv x 4 extension values (one row per interaction) and niv
// more sumcheck rounds over them (`InteractionLayer`, /root/reference/crates/hypercube/src/logup_gkr/logup_poly.rs:L240-L316 as restated in
// gkr.hip). They run on the host because each is a hand-over of its own and the tables are tiny — but at 730 interactions
// the scalar form is ~10k extension products per layer, 57 us with eight helper threads forking and joining twenty
// times, while the GPU waits (27 layers per proof). Here eight pairs go through one set of AVX-512 registers: the tables
// are kept as coefficient planes (structure of arrays), a 512-bit load of 16 consecutive words IS eight (even, odd)
// pairs in the 64-bit lanes vpmuludq wants, an extension product is 16 vpmuludq accumulated per coefficient in 64 bits
// (operands of the x^4 = 3 wrap pre-tripled, one wide Montgomery reduction per coefficient, like kb::ext_mul on the GPU).
// One thread, no fork / join: ~10 us per layer. Same field elements as the scalar code in gkr.hip, which stays as the
// path for CPUs without AVX-512 (and under SP1HIP_HOST_SIMD=0; the GPU tests run both).
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kb31.hpp"

namespace sp1hip {

#define GKRH_TARGET __attribute__((target("avx512f,avx512dq,avx512vl,avx512bw")))

namespace {

struct V4 { __m512i c[4]; };           // eight extension elements, coefficient k of element l in 64-bit lane l of c[k]

GKRH_TARGET inline __m512i vp() { return _mm512_set1_epi64(kb::P); }
GKRH_TARGET inline __m512i fadd(__m512i a, __m512i b) { const __m512i s = _mm512_add_epi32(a, b); return _mm512_min_epu32(s, _mm512_sub_epi32(s, vp())); }
GKRH_TARGET inline __m512i fsub(__m512i a, __m512i b) { const __m512i d = _mm512_sub_epi32(a, b); return _mm512_min_epu32(d, _mm512_add_epi32(d, vp())); }
// NOTE on fadd / fsub: the lanes are 64 bits wide with the value in the low word and a zero high word; the 32-bit
// operations leave the high word zero (0 + 0, 0 - 0, and min(0, 0 - 0 = 0)); fsub's wrap-around stays inside the low word.

GKRH_TARGET inline V4 vadd(const V4& a, const V4& b) { V4 r; for (int k = 0; k < 4; k++) r.c[k] = fadd(a.c[k], b.c[k]); return r; }
GKRH_TARGET inline V4 vsub(const V4& a, const V4& b) { V4 r; for (int k = 0; k < 4; k++) r.c[k] = fsub(a.c[k], b.c[k]); return r; }
GKRH_TARGET inline __m512i triple(__m512i x) { return fadd(fadd(x, x), x); }

// x < 4 p^2 (< 2^64) per lane -> x 2^-32 mod p, canonical. The high word is < 2 p: one conditional subtraction brings x
// under 2^32 p, then the additive Montgomery step (x + (lo(x) * -p^-1 mod 2^32) p) >> 32 < 2 p and one correction.
GKRH_TARGET inline __m512i reduce_wide(__m512i x) {
    const __m512i p = vp(), nmu = _mm512_set1_epi64(kb::NMU);
    const __m512i hi = _mm512_srli_epi64(x, 32);
    const __m512i d = _mm512_sub_epi32(hi, _mm512_min_epu32(hi, _mm512_sub_epi32(hi, p)));        // 0 or p
    x = _mm512_sub_epi64(x, _mm512_slli_epi64(d, 32));
    const __m512i t = _mm512_mul_epu32(x, nmu);
    const __m512i r = _mm512_srli_epi64(_mm512_add_epi64(_mm512_mul_epu32(t, p), x), 32);
    return _mm512_min_epu32(r, _mm512_sub_epi32(r, p));
}

struct V4T { V4 v; __m512i t1, t2, t3; };         // an operand with 3 b1, 3 b2, 3 b3 beside it
GKRH_TARGET inline V4T with_triples(const V4& b) { return V4T{b, triple(b.c[1]), triple(b.c[2]), triple(b.c[3])}; }

#define MUL(x, y) _mm512_mul_epu32(x, y)
#define ADD(x, y) _mm512_add_epi64(x, y)
GKRH_TARGET inline V4 vmul(const V4& a, const V4T& b) {
    V4 r;
    r.c[0] = reduce_wide(ADD(ADD(MUL(a.c[0], b.v.c[0]), MUL(a.c[1], b.t3)), ADD(MUL(a.c[2], b.t2), MUL(a.c[3], b.t1))));
    r.c[1] = reduce_wide(ADD(ADD(MUL(a.c[0], b.v.c[1]), MUL(a.c[1], b.v.c[0])), ADD(MUL(a.c[2], b.t3), MUL(a.c[3], b.t2))));
    r.c[2] = reduce_wide(ADD(ADD(MUL(a.c[0], b.v.c[2]), MUL(a.c[1], b.v.c[1])), ADD(MUL(a.c[2], b.v.c[0]), MUL(a.c[3], b.t3))));
    r.c[3] = reduce_wide(ADD(ADD(MUL(a.c[0], b.v.c[3]), MUL(a.c[1], b.v.c[2])), ADD(MUL(a.c[2], b.v.c[1]), MUL(a.c[3], b.v.c[0]))));
    return r;
}
// a b + c d with ONE reduction per coefficient would be 8 products < 8 p^2: does not fit 64 bits; two products it is.
#undef MUL
#undef ADD

// eight (even, odd) pairs of one table: planes[k] + 2 k0 .. + 16
GKRH_TARGET inline void load_pairs(const uint32_t* plane0, size_t stride, size_t k0, V4& even, V4& odd) {
    const __m512i lo = _mm512_set1_epi64(0xffffffffll);
    for (int k = 0; k < 4; k++) {
        const __m512i x = _mm512_loadu_si512(plane0 + (size_t)k * stride + 2 * k0);
        even.c[k] = _mm512_and_si512(x, lo);
        odd.c[k] = _mm512_srli_epi64(x, 32);
    }
}

GKRH_TARGET inline void hsum(const V4& a, uint32_t out[4]) {
    alignas(64) uint64_t l[8];
    for (int k = 0; k < 4; k++) {
        _mm512_store_si512(l, a.c[k]);
        uint64_t s = 0;
        for (int i = 0; i < 8; i++) s += l[i];            // 8 canonical words: < 2^35
        out[k] = (uint32_t)(s % kb::P);
    }
}

GKRH_TARGET void sums_avx512(const uint32_t* tab, size_t stride, const uint32_t* eq, size_t eq_stride, size_t real_pairs, uint32_t out[6][4]) {
    V4 x0{}, y0{}, xh{}, yh{}, e0{}, es{};
    for (int k = 0; k < 4; k++) x0.c[k] = y0.c[k] = xh.c[k] = yh.c[k] = e0.c[k] = es.c[k] = _mm512_setzero_si512();
    const uint32_t *n0p = tab, *d0p = tab + 4 * stride, *n1p = tab + 8 * stride, *d1p = tab + 12 * stride;
    for (size_t k0 = 0; k0 < real_pairs; k0 += 8) {
        const __mmask8 live = real_pairs - k0 >= 8 ? (__mmask8)0xff : (__mmask8)((1u << (real_pairs - k0)) - 1u);
        V4 n0a, n0b, d0a, d0b, n1a, n1b, d1a, d1b, ea, eb;
        load_pairs(n0p, stride, k0, n0a, n0b);
        load_pairs(d0p, stride, k0, d0a, d0b);
        load_pairs(n1p, stride, k0, n1a, n1b);
        load_pairs(d1p, stride, k0, d1a, d1b);
        load_pairs(eq, eq_stride, k0, ea, eb);
        for (int k = 0; k < 4; k++) {                       // lanes past the last real pair contribute nothing
            ea.c[k] = _mm512_maskz_mov_epi64(live, ea.c[k]);
            eb.c[k] = _mm512_maskz_mov_epi64(live, eb.c[k]);
        }
        // at 0: eq[a] (d0 n1 + d1 n0), eq[a] d0 d1
        {
            const V4T n1t = with_triples(n1a), n0t = with_triples(n0a), d1t = with_triples(d1a), eat = with_triples(ea);
            const V4 X = vadd(vmul(d0a, n1t), vmul(d1a, n0t));
            const V4 Y = vmul(d0a, d1t);
            x0 = vadd(x0, vmul(X, eat));
            y0 = vadd(y0, vmul(Y, eat));
            e0 = vadd(e0, ea);
        }
        // at "1/2" (unscaled): (eq[a] + eq[b]) ((d0a + d0b)(n1a + n1b) + (d1a + d1b)(n0a + n0b)), ... (d0a + d0b)(d1a + d1b)
        {
            const V4 sn0 = vadd(n0a, n0b), sn1 = vadd(n1a, n1b), sd0 = vadd(d0a, d0b), sd1 = vadd(d1a, d1b), e = vadd(ea, eb);
            const V4T sn1t = with_triples(sn1), sn0t = with_triples(sn0), sd1t = with_triples(sd1), et = with_triples(e);
            const V4 X = vadd(vmul(sd0, sn1t), vmul(sd1, sn0t));
            const V4 Y = vmul(sd0, sd1t);
            xh = vadd(xh, vmul(X, et));
            yh = vadd(yh, vmul(Y, et));
            es = vadd(es, e);
        }
    }
    hsum(x0, out[0]); hsum(y0, out[1]); hsum(xh, out[2]); hsum(yh, out[3]); hsum(e0, out[4]); hsum(es, out[5]);
}

GKRH_TARGET void fold_avx512(const uint32_t* tab, uint32_t* out, size_t stride, size_t real_pairs, const uint32_t alpha[4]) {
    V4 a;
    for (int k = 0; k < 4; k++) a.c[k] = _mm512_set1_epi64(alpha[k]);
    const V4T at = with_triples(a);
    for (int w = 0; w < 4; w++) {
        const uint32_t* src = tab + (size_t)w * 4 * stride;
        uint32_t* dst = out + (size_t)w * 4 * stride;
        for (size_t k0 = 0; k0 < real_pairs; k0 += 8) {
            V4 ev, od;
            load_pairs(src, stride, k0, ev, od);
            const V4 r = vadd(ev, vmul(vsub(od, ev), at));
            for (int k = 0; k < 4; k++) _mm256_storeu_si256((__m256i*)(dst + (size_t)k * stride + k0), _mm512_cvtepi64_epi32(r.c[k]));
        }
    }
}

bool simd_ok() {
    static const bool cpu = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
                            __builtin_cpu_supports("avx512bw");
    const char* e = getenv("SP1HIP_HOST_SIMD");           // read per call: the GPU tests prove with both paths in one process
    return cpu && !(e && atoi(e) == 0);
}

}  // namespace

// Interface for gkr.hip. Tables: 4 tables (n0, d0, n1, d1) x 4 coefficient planes of `stride` words each, table w plane k at
// tab + (4 w + k) stride; stride is a multiple of 16 and at least 2 (pairs rounded up to 8) + 16 so that whole vectors can
// be read and written past the live entries. eq: 4 planes of eq_stride words.
bool gkr_host_simd_available() { return simd_ok(); }

void gkr_host_round_sums(const uint32_t* tab, size_t stride, const uint32_t* eq, size_t eq_stride, size_t real_pairs, uint32_t out[6][4]) {
    sums_avx512(tab, stride, eq, eq_stride, real_pairs, out);
}

void gkr_host_round_fold(const uint32_t* tab, uint32_t* out, size_t stride, size_t real_pairs, const uint32_t alpha[4]) {
    fold_avx512(tab, out, stride, real_pairs, alpha);
}

}  // namespace sp1hip

// Host-only test hooks (include/sp1hip.h): the two primitives above on caller-provided planes.
extern "C" {
int sp1hip_gkr_host_simd_available(void) { return sp1hip::gkr_host_simd_available() ? 1 : 0; }
int sp1hip_gkr_host_round_sums(const uint32_t* tab, size_t stride, const uint32_t* eq, size_t eq_stride, size_t real_pairs, uint32_t* out24) {
    if (!sp1hip::gkr_host_simd_available() || !tab || !eq || !out24 || stride % 16 || eq_stride % 16) return -1;
    uint32_t o[6][4];
    sp1hip::gkr_host_round_sums(tab, stride, eq, eq_stride, real_pairs, o);
    memcpy(out24, o, sizeof o);
    return 0;
}
int sp1hip_gkr_host_round_fold(const uint32_t* tab, uint32_t* out, size_t stride, size_t real_pairs, const uint32_t* alpha4) {
    if (!sp1hip::gkr_host_simd_available() || !tab || !out || !alpha4 || stride % 16) return -1;
    sp1hip::gkr_host_round_fold(tab, out, stride, real_pairs, alpha4);
    return 0;
}
}
