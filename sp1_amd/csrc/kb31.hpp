// sp1_amd/csrc/kb31.hpp — KoalaBear (p = 2^31 - 2^24 + 1) arithmetic for gfx950 device code and the
// host-side transcript. Words are Montgomery form, R = 2^32, exactly the in-memory representation of
// the reference's `SP1Field` (/root/reference/crates/primitives/src/lib.rs:L28-L31; Montgomery
// conventions as restated in /root/reference/sp1-gpu/crates/sys/include/fields/kb31_t.cuh:L70-L135).
//
// gfx950 notes (measured in profiles/r01_ubench.txt): v_mul_lo/hi_u32 issue
// at about the same rate as other VOP3 integer ops and v_mad_u64_u32 at ~1.3x that, so the kernels are
// bound by VALU *instruction count*, not by a slow multiplier:
//  * Montgomery reduction uses the additive form (one v_mul_lo_u32 + one v_mad_u64_u32),
//  * conditional corrections use the unsigned-min idiom (v_min_u32) instead of compare/select,
//  * corrections are skipped wherever the next consumer tolerates a value in [0, 2p),
//  * extension-field products accumulate pairs of 64-bit products before reducing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KB_HD __host__ __device__ __forceinline__

namespace kb {

constexpr uint32_t P = 0x7f000001u;
constexpr uint32_t MU = 0x81000001u;       // p^-1 mod 2^32 = 2^31 + 2^24 + 1
constexpr uint32_t NMU = 0x7effffffu;      // -p^-1 mod 2^32
constexpr uint32_t R1 = 0x01fffffeu;       // 2^32 mod p   (Montgomery one)
constexpr uint32_t R2 = 0x17f7efe4u;       // 2^64 mod p
constexpr uint32_t GEN24 = 0x6ac49f88u;    // canonical two_adic_generator(24)
constexpr int TWO_ADICITY = 24;

KB_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

KB_HD uint32_t mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// Montgomery reduction, "plus" form: with t = x_lo * (-p^-1) mod 2^32 the sum x + t p is a multiple of
// 2^32, so (x + t p) >> 32 is x * 2^-32 (mod p) and lies in [x / 2^32, x / 2^32 + p). Needs x < 2^63.
// On gfx950 this is v_mul_lo_u32 + one v_mad_u64_u32 (the multiply-add swallows the accumulation);
// measured 12-17 % faster than the subtractive form (bench/ubench_*.hip, profiles/r01_ubench.txt).
KB_HD uint32_t monty_reduce_lazy(uint64_t x) {
    uint32_t t = (uint32_t)x * NMU;
    return (uint32_t)(((uint64_t)t * P + x) >> 32);
}

// x < 2^32 * p  ->  x * 2^-32 mod p, fully reduced
KB_HD uint32_t monty_reduce(uint64_t x) {
    uint32_t r = monty_reduce_lazy(x);   // in [0, 2p)
    return umin(r, r - P);
}

KB_HD uint32_t add(uint32_t a, uint32_t b) { uint32_t s = a + b; return umin(s, s - P); }
KB_HD uint32_t sub(uint32_t a, uint32_t b) { uint32_t d = a - b; return umin(d, d + P); }
KB_HD uint32_t neg(uint32_t a) { return sub(0u, a); }
KB_HD uint32_t dbl(uint32_t a) { return add(a, a); }
KB_HD uint32_t mul(uint32_t a, uint32_t b) { return monty_reduce((uint64_t)a * b); }
KB_HD uint32_t sqr(uint32_t a) { return mul(a, a); }
KB_HD uint32_t to_monty(uint32_t canonical) { return mul(canonical, R2); }
KB_HD uint32_t from_monty(uint32_t m) { return monty_reduce((uint64_t)m); }

KB_HD uint32_t pow(uint32_t b, uint64_t e) {
    uint32_t r = R1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return r;
}
KB_HD uint32_t inv(uint32_t a) { return pow(a, P - 2); }

KB_HD uint32_t two_adic_generator(int bits) {
    uint32_t g = to_monty(GEN24);
    for (int i = bits; i < TWO_ADICITY; i++) g = sqr(g);
    return g;
}

KB_HD uint32_t reverse_bits_len(uint32_t x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0u;
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

// ---- EF = F[x]/(x^4 - 3); c[0..3] = coefficient order = base-slice order ------------------------
struct Ext {
    uint32_t c[4];
};

KB_HD Ext ext_zero() { return Ext{{0, 0, 0, 0}}; }
KB_HD Ext ext_one() { return Ext{{R1, 0, 0, 0}}; }
KB_HD Ext ext_from_base(uint32_t b) { return Ext{{b, 0, 0, 0}}; }
KB_HD bool ext_eq(const Ext& a, const Ext& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
KB_HD Ext ext_add(const Ext& a, const Ext& b) {
    return Ext{{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}};
}
KB_HD Ext ext_sub(const Ext& a, const Ext& b) {
    return Ext{{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}};
}
KB_HD Ext ext_mul_base(const Ext& a, uint32_t b) {
    return Ext{{mul(a.c[0], b), mul(a.c[1], b), mul(a.c[2], b), mul(a.c[3], b)}};
}

// x < 2^64 - 2^58 (any sum of four products of reduced words: 4 (p - 1)^2 = 2^64 - 2^58 + 2^50)  ->  x * 2^-32 mod p, fully
// reduced. The high word is < 2^32 - 2^26 + 2^18 < 2p, so one conditional subtraction brings x under 2^32 p, where the
// additive reduction applies: 6 VALU instructions (sub, min, mul_lo, mad_u64, sub, min) for FOUR products instead of two
// reductions of two products and a modular add (11).
KB_HD uint32_t monty_reduce_wide(uint64_t x) {
    const uint32_t h = (uint32_t)(x >> 32);
    return monty_reduce(((uint64_t)umin(h, h - P) << 32) | (uint32_t)x);
}

// Operand order matters for speed only: the x^4 = 3 wrap multiples are formed from the SECOND operand, so pass the
// wave-uniform factor (a challenge, a power table entry) second — its multiples are then computed once on the scalar
// unit / hoisted out of the loop instead of 18 VALU instructions per product.
// Full product with delayed reduction: every output coefficient is ONE 64-bit accumulation of its four products (the
// products ride v_mad_u64_u32) and one reduction (monty_reduce_wide). The x^4 = 3 wrap is applied to b up front with two
// modular additions per coefficient instead of a multiply. 16 wide multiply-adds + 24 instructions of reduction (+ 18
// for the wrap multiples of a non-uniform b): 58 VALU instructions, was 78 with a reduction per pair of products.
KB_HD Ext ext_mul(const Ext& a, const Ext& b) {
    const uint32_t w1 = add(dbl(b.c[1]), b.c[1]), w2 = add(dbl(b.c[2]), b.c[2]), w3 = add(dbl(b.c[3]), b.c[3]);
    Ext r;
    r.c[0] = monty_reduce_wide((uint64_t)a.c[0] * b.c[0] + (uint64_t)a.c[1] * w3 + (uint64_t)a.c[2] * w2 + (uint64_t)a.c[3] * w1);
    r.c[1] = monty_reduce_wide((uint64_t)a.c[0] * b.c[1] + (uint64_t)a.c[1] * b.c[0] + (uint64_t)a.c[2] * w3 + (uint64_t)a.c[3] * w2);
    r.c[2] = monty_reduce_wide((uint64_t)a.c[0] * b.c[2] + (uint64_t)a.c[1] * b.c[1] + (uint64_t)a.c[2] * b.c[0] + (uint64_t)a.c[3] * w3);
    r.c[3] = monty_reduce_wide((uint64_t)a.c[0] * b.c[3] + (uint64_t)a.c[1] * b.c[2] + (uint64_t)a.c[2] * b.c[1] + (uint64_t)a.c[3] * b.c[0]);
    return r;
}

// ---- long dot products sum_r e_r * v_r (e extension, v base) with ONE reduction at the end ------------------------
// v is split into 16-bit halves so that a partial product e_k * half fits 47 bits and 2^16 of them fit a 64-bit
// accumulator: per term and coordinate two v_mad_u64_u32 and no reduction, instead of a 64-bit product, a Montgomery
// reduction and a modular add (~15 VALU slots per term against ~38). Used by the evaluation kernels (columns at a
// point, batching): their inner loops are exactly this shape.
struct DotAcc { uint64_t lo[4], hi[4]; };
KB_HD void dot_init(DotAcc& a) {
    for (int k = 0; k < 4; k++) { a.lo[k] = 0; a.hi[k] = 0; }
}
KB_HD void dot_add(DotAcc& a, const Ext& e, uint32_t v) {      // at most 2^16 calls between dot_init and dot_finish
    const uint32_t vl = v & 0xffffu, vh = v >> 16;
    for (int k = 0; k < 4; k++) {
        a.lo[k] += (uint64_t)e.c[k] * vl;
        a.hi[k] += (uint64_t)e.c[k] * vh;
    }
}
// S < 2^63 -> S * 2^-32 mod p (Montgomery reduction of the low word; the high word is already "times 2^32")
KB_HD uint32_t dot_reduce64(uint64_t s) {
    const uint32_t h = (uint32_t)(s >> 32);                     // < 2^31 < 2p
    return add(monty_reduce((uint64_t)(uint32_t)s), umin(h, h - P));
}
KB_HD Ext dot_finish(const DotAcc& a) {
    constexpr uint32_t M16 = (uint32_t)(((uint64_t)1 << 48) % P);   // to_monty(2^16)
    Ext r;
    for (int k = 0; k < 4; k++) r.c[k] = add(dot_reduce64(a.lo[k]), mul(dot_reduce64(a.hi[k]), M16));
    return r;
}

// The same for extension x extension terms sum_r a_r * b_r when a_r is known together with 3 a_r (the x^4 = 3 wrap):
// every output coordinate is four products per term; b's words are split into 16-bit halves, so a term is 32
// multiply-adds and no reduction (~59 VALU slots against ~104 for ext_mul + ext_add). At most 2^14 terms.
KB_HD void edot_add(DotAcc& acc, const Ext& a, const Ext& a3, const Ext& b) {
    uint32_t bl[4], bh[4];
    for (int j = 0; j < 4; j++) { bl[j] = b.c[j] & 0xffffu; bh[j] = b.c[j] >> 16; }
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 4; i++) {
            const int j = (k - i) & 3;
            const uint32_t coef = (i + j == k) ? a.c[i] : a3.c[i];      // i + j == k + 4: wrapped, times 3
            acc.lo[k] += (uint64_t)coef * bl[j];
            acc.hi[k] += (uint64_t)coef * bh[j];
        }
}

// Inverse through the tower F < F[y]/(y^2-3) < EF, y = x^2 (host-side transcript use only).
KB_HD Ext ext_inv(const Ext& a) {
    const uint32_t W = 0x05fffffau;
    uint32_t A0 = a.c[0], A1 = a.c[2], B0 = a.c[1], B1 = a.c[3];
    uint32_t A2_0 = add(sqr(A0), mul(W, sqr(A1))), A2_1 = dbl(mul(A0, A1));
    uint32_t B2_0 = add(sqr(B0), mul(W, sqr(B1))), B2_1 = dbl(mul(B0, B1));
    uint32_t D0 = sub(A2_0, mul(W, B2_1)), D1 = sub(A2_1, B2_0);
    uint32_t n = inv(sub(sqr(D0), mul(W, sqr(D1))));
    uint32_t I0 = mul(D0, n), I1 = neg(mul(D1, n));
    uint32_t rA0 = add(mul(A0, I0), mul(W, mul(A1, I1))), rA1 = add(mul(A0, I1), mul(A1, I0));
    uint32_t rB0 = add(mul(B0, I0), mul(W, mul(B1, I1))), rB1 = add(mul(B0, I1), mul(B1, I0));
    return Ext{{rA0, neg(rB0), rA1, neg(rB1)}};
}

}  // namespace kb
