// Modular arithmetic on up to 384-bit little-endian u64 limbs for the executor's field / curve system calls (rv64_exec.cpp):
// what `num::BigUint` does for the reference's precompile handlers
// (/root/reference/crates/core/executor/src/minimal/precompiles/{ec.rs,fptower/*.rs,edwards/*.rs}). Host code only; a few
// thousand calls per program, so one generic Montgomery form (CIOS) over any odd modulus instead of a special reduction per field.
#pragma once
#include <cstdint>
#include <cstring>

namespace bigmod {

constexpr int MAXW = 6;
struct Int { uint64_t w[MAXW] = {0, 0, 0, 0, 0, 0}; };

inline bool is_zero(const Int& a) { uint64_t o = 0; for (int k = 0; k < MAXW; ++k) o |= a.w[k]; return o == 0; }
inline bool eq(const Int& a, const Int& b) { return memcmp(a.w, b.w, sizeof a.w) == 0; }
inline bool ge(const Int& a, const Int& b) { for (int k = MAXW - 1; k >= 0; --k) if (a.w[k] != b.w[k]) return a.w[k] > b.w[k]; return true; }
inline uint64_t add_into(Int& r, const Int& a, const Int& b) {        // returns the carry out of limb MAXW - 1
    unsigned __int128 c = 0;
    for (int k = 0; k < MAXW; ++k) { c += (unsigned __int128)a.w[k] + b.w[k]; r.w[k] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
inline uint64_t sub_into(Int& r, const Int& a, const Int& b) {        // returns the borrow
    unsigned __int128 borrow = 0;
    for (int k = 0; k < MAXW; ++k) { const unsigned __int128 t = (unsigned __int128)a.w[k] - b.w[k] - borrow; r.w[k] = (uint64_t)t; borrow = (t >> 64) & 1; }
    return (uint64_t)borrow;
}
inline Int from_words(const uint64_t* w, int n) { Int r; for (int k = 0; k < n; ++k) r.w[k] = w[k]; return r; }
inline Int small(uint64_t v) { Int r; r.w[0] = v; return r; }

struct Field {
    int n = 0;                                                         // limbs in use (4 or 6); R = 2^(64 n)
    Int p, r2, one_m;                                                  // modulus, R^2 mod p, R mod p
    uint64_t n0 = 0;                                                   // -p^{-1} mod 2^64

    Field() {}
    Field(const uint64_t* words, int limbs) : n(limbs) {
        p = from_words(words, limbs);
        uint64_t inv = 1;                                              // Newton: inv = p^{-1} mod 2^64 (p odd)
        for (int i = 0; i < 6; ++i) inv *= 2 - p.w[0] * inv;
        n0 = 0 - inv;
        Int r = small(1);
        for (int i = 0; i < 128 * limbs; ++i) {                        // 2^(128 n) mod p by doubling; R mod p on the way
            if (i == 64 * limbs) one_m = r;
            r = dbl(r);
        }
        r2 = r;
    }
    Int dbl(const Int& a) const { return add(a, a); }
    Int add(const Int& a, const Int& b) const {                        // a, b < p < 2^(64 n): no carry out of MAXW limbs unless n = MAXW
        Int r; const uint64_t c = add_into(r, a, b);
        if (c || ge(r, p)) { Int t; sub_into(t, r, p); return t; }
        return r;
    }
    Int sub(const Int& a, const Int& b) const {
        Int r; if (sub_into(r, a, b)) { Int t; add_into(t, r, p); return t; }
        return r;
    }
    Int neg(const Int& a) const { return is_zero(a) ? a : sub(p, a); }
    Int montmul(const Int& a, const Int& b) const {                    // a b / R mod p (coarsely integrated operand scanning)
        uint64_t t[MAXW + 2] = {0};
        for (int i = 0; i < n; ++i) {
            unsigned __int128 c = 0;
            for (int j = 0; j < n; ++j) { c += (unsigned __int128)a.w[j] * b.w[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[n]; t[n] = (uint64_t)c; t[n + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * n0;
            c = (unsigned __int128)m * p.w[0] + t[0]; c >>= 64;
            for (int j = 1; j < n; ++j) { c += (unsigned __int128)m * p.w[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[n]; t[n - 1] = (uint64_t)c; t[n] = t[n + 1] + (uint64_t)(c >> 64);
        }
        Int r = from_words(t, n);
        if (t[n] || ge(r, p)) { Int s; sub_into(s, r, p); for (int k = n; k < MAXW; ++k) s.w[k] = 0; return s; }
        return r;
    }
    Int to_m(const Int& a) const { return montmul(a, r2); }
    Int from_m(const Int& a) const { return montmul(a, small(1)); }
    Int mul(const Int& a, const Int& b) const { return montmul(montmul(a, b), r2); }
    Int pow(const Int& a, const Int& e) const {                        // a^e, plain in and out
        Int base = to_m(a), r = one_m;
        for (int bit = 0; bit < 64 * n; ++bit) { if ((e.w[bit >> 6] >> (bit & 63)) & 1) r = montmul(r, base); base = montmul(base, base); }
        return from_m(r);
    }
    Int inv(const Int& a) const { Int e; sub_into(e, p, small(2)); return pow(a, e); }   // Fermat: p prime
    bool reduced(const Int& a) const { return !ge(a, p); }
};

}  // namespace bigmod
