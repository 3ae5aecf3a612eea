// oracle/kb_field.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// KoalaBear base field and its degree-4 binomial extension, restated for the CPU from the
// reference. All in-memory words are Montgomery form with R = 2^32, exactly as the reference keeps
// them (serde/bincode uses canonical words).
//
//   p = 2^31 - 2^24 + 1 = 0x7f000001, two-adicity 24, EF = F[x]/(x^4 - 3)
//     /root/reference/crates/primitives/src/lib.rs:L28-L38
//   Montgomery conventions (MU = p^-1 mod 2^32, reduce = (x - (x*MU mod 2^32)*p) >> 32, +p on borrow)
//     /root/reference/sp1-gpu/crates/sys/include/fields/kb31_t.cuh:L70-L135
//   generator 3, two_adic_generator(24) = 0x6ac49f88 (canonical)
//     /root/reference/sp1-gpu/crates/sys/sppark/ntt/parameters/koala_bear.h:L5-L36
//
// The arithmetic itself lives in the un-vendored dependency Plonky3 `p3-koala-bear`/`p3-field`
// `=0.4.3-succinct` (/root/reference/Cargo.lock:L4649-L4870); this file restates the published
// algorithm. Pinned by tests/golden/kb_shrink_basefold.npz (real reference proof data).
#pragma once
#include <cstdint>
#include <cstring>

namespace orc {

constexpr uint32_t KB_P = 0x7f000001u;
constexpr uint32_t KB_MU = 0x81000001u;        // p^-1 mod 2^32
constexpr int KB_TWO_ADICITY = 24;
constexpr uint32_t KB_GEN24_CANON = 0x6ac49f88u;
constexpr uint32_t KB_EXT_W_CANON = 3;

static inline uint32_t monty_reduce(uint64_t x) {
    uint32_t t = (uint32_t)x * KB_MU;
    uint64_t u = (uint64_t)t * KB_P;
    uint64_t d = x - u;
    uint32_t hi = (uint32_t)(d >> 32);
    return x < u ? hi + KB_P : hi;
}

struct F {
    uint32_t v;  // Montgomery word, always < p
    static F raw(uint32_t w) { F r; r.v = w; return r; }
    static F zero() { return raw(0); }
    static F from_canonical(uint32_t c) {
        // R^2 mod p computed once
        static const uint32_t R2 = []() {
            uint64_t r = ((uint64_t)1 << 32) % KB_P;
            return (uint32_t)((r * r) % KB_P);
        }();
        return raw(monty_reduce((uint64_t)(c % KB_P) * R2));
    }
    static F one() { static const F o = from_canonical(1); return o; }
    static F two() { static const F o = from_canonical(2); return o; }
    uint32_t canonical() const { return monty_reduce(v); }
    bool operator==(const F& o) const { return v == o.v; }
    bool operator!=(const F& o) const { return v != o.v; }
    bool is_zero() const { return v == 0; }
};

static inline F operator+(F a, F b) { uint32_t s = a.v + b.v; return F::raw(s >= KB_P ? s - KB_P : s); }
static inline F operator-(F a, F b) { return F::raw(a.v >= b.v ? a.v - b.v : a.v + KB_P - b.v); }
static inline F operator-(F a) { return F::raw(a.v ? KB_P - a.v : 0); }
static inline F operator*(F a, F b) { return F::raw(monty_reduce((uint64_t)a.v * b.v)); }
static inline F& operator+=(F& a, F b) { a = a + b; return a; }
static inline F& operator-=(F& a, F b) { a = a - b; return a; }
static inline F& operator*=(F& a, F b) { a = a * b; return a; }

static inline F fpow(F b, uint64_t e) {
    F r = F::one();
    while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
    return r;
}
static inline F finv(F a) { return fpow(a, KB_P - 2); }

static inline F two_adic_generator(int bits) {
    F g = F::from_canonical(KB_GEN24_CANON);
    for (int i = bits; i < KB_TWO_ADICITY; i++) g *= g;
    return g;
}

// ---- EF = F[x]/(x^4 - 3); coefficient order = base-slice order ---------------------------------
struct E {
    F c[4];
    static E zero() { E e; for (auto& x : e.c) x = F::zero(); return e; }
    static E one() { E e = zero(); e.c[0] = F::one(); return e; }
    static E from_base(F b) { E e = zero(); e.c[0] = b; return e; }
    bool operator==(const E& o) const { return !memcmp(c, o.c, sizeof c); }
    bool operator!=(const E& o) const { return !(*this == o); }
};
static inline F ext_w() { static const F w = F::from_canonical(KB_EXT_W_CANON); return w; }

static inline E operator+(E a, const E& b) { for (int i = 0; i < 4; i++) a.c[i] += b.c[i]; return a; }
static inline E operator-(E a, const E& b) { for (int i = 0; i < 4; i++) a.c[i] -= b.c[i]; return a; }
static inline E operator-(E a) { for (int i = 0; i < 4; i++) a.c[i] = -a.c[i]; return a; }
static inline E operator*(const E& a, F b) { E r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] * b; return r; }
static inline E operator+(E a, F b) { a.c[0] += b; return a; }
static inline E operator-(E a, F b) { a.c[0] -= b; return a; }
static inline E operator*(const E& a, const E& b) {
    const F w = ext_w();
    E r;
    r.c[0] = a.c[0] * b.c[0] + w * (a.c[1] * b.c[3] + a.c[2] * b.c[2] + a.c[3] * b.c[1]);
    r.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0] + w * (a.c[2] * b.c[3] + a.c[3] * b.c[2]);
    r.c[2] = a.c[0] * b.c[2] + a.c[1] * b.c[1] + a.c[2] * b.c[0] + w * (a.c[3] * b.c[3]);
    r.c[3] = a.c[0] * b.c[3] + a.c[1] * b.c[2] + a.c[2] * b.c[1] + a.c[3] * b.c[0];
    return r;
}
static inline E& operator+=(E& a, const E& b) { a = a + b; return a; }
static inline E& operator-=(E& a, const E& b) { a = a - b; return a; }
static inline E& operator*=(E& a, const E& b) { a = a * b; return a; }

// Inverse through the tower F ⊂ F[y]/(y^2-3) ⊂ EF with y = x^2:
// a = A + xB, A = a0 + a2 y, B = a1 + a3 y;  a^-1 = (A - xB) / (A^2 - y B^2).
static inline E einv(const E& a) {
    const F w = ext_w();
    F A0 = a.c[0], A1 = a.c[2], B0 = a.c[1], B1 = a.c[3];
    // A^2 = (A0^2 + w A1^2) + (2 A0 A1) y ; B^2 likewise ; y*B^2 = w*B2_1 + B2_0 y
    F A2_0 = A0 * A0 + w * A1 * A1, A2_1 = F::two() * A0 * A1;
    F B2_0 = B0 * B0 + w * B1 * B1, B2_1 = F::two() * B0 * B1;
    F D0 = A2_0 - w * B2_1, D1 = A2_1 - B2_0;
    F n = finv(D0 * D0 - w * D1 * D1);
    F I0 = D0 * n, I1 = -(D1 * n);                      // D^-1 = I0 + I1 y
    // (A - xB) * D^-1
    F rA0 = A0 * I0 + w * A1 * I1, rA1 = A0 * I1 + A1 * I0;
    F rB0 = B0 * I0 + w * B1 * I1, rB1 = B0 * I1 + B1 * I0;
    E r;
    r.c[0] = rA0; r.c[2] = rA1; r.c[1] = -rB0; r.c[3] = -rB1;
    return r;
}

static inline uint32_t reverse_bits_len(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

}  // namespace orc
