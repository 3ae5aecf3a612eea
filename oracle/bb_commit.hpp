// oracle/bb_commit.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// BabyBear instantiation of the commit path (BASELINE config 2 names it: "NTT/LDE + Poseidon2 commit only ... both fields"):
// field, RS encode and Poseidon2 Merkle commitment restated for the CPU, the same algorithms as kb_field.hpp / kb_hash.hpp /
// kb_pcs.hpp with BabyBear's parameters:
//   p = 2^31 - 2^27 + 1 = 0x78000001, two-adicity 27, multiplicative generator 31 (published Plonky3 p3-baby-bear constants;
//     the reference uses the crate through /root/reference/slop/crates/baby-bear/src/lib.rs)
//   Poseidon2 width 16, S-box x^7, 8 external + 13 internal rounds, round constants RC_16_30
//     /root/reference/slop/crates/baby-bear/src/baby_bear_poseidon2.rs:L11-L30,L57-...
//   sponge / compression / commit_tensors exactly as for KoalaBear
//     /root/reference/slop/crates/baby-bear/src/baby_bear_poseidon2.rs:L36-L55, slop/crates/merkle-tree/src/p3sync.rs:L40-L143
//
// **PARITY UNPINNED** for one parameter: the internal (diffusion) matrix `DiffusionMatrixBabyBear` lives in the un-vendored
// dependency p3-baby-bear =0.4.3-succinct and nothing in the reference tree restates it or holds BabyBear hash outputs. Used
// here, as SURVEY §8c prescribes: the published Plonky3 convention of that version, s_i <- (sum + d_i s_i) 2^-32 with
// d = [-2, 1, 2, 4, ..., 2^13, 2^15] — the same shape the tree does pin for KoalaBear (kb_hash.hpp). The field, the NTT, the
// external layer, the S-box, the round constants and the sponge are pinned by the reference tree / published constants.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orcbb {

constexpr uint32_t P = 0x78000001u;
constexpr int TWO_ADICITY = 27;

static inline uint32_t mu() {                    // p^-1 mod 2^32 (Newton)
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - P * x;
    return x;
}
static inline uint32_t monty_reduce(uint64_t x) {
    static const uint32_t MU = mu();
    uint32_t t = (uint32_t)x * MU;
    uint64_t u = (uint64_t)t * P;
    uint64_t d = x - u;
    uint32_t hi = (uint32_t)(d >> 32);
    return x < u ? hi + P : hi;
}
struct F {
    uint32_t v;                                   // Montgomery word, R = 2^32
    static F raw(uint32_t w) { F r; r.v = w; return r; }
    static F zero() { return raw(0); }
    static F from_canonical(uint32_t c) {
        static const uint32_t R2 = []() { uint64_t r = ((uint64_t)1 << 32) % P; return (uint32_t)((r * r) % P); }();
        return raw(monty_reduce((uint64_t)(c % P) * R2));
    }
    static F one() { static const F o = from_canonical(1); return o; }
    uint32_t canonical() const { return monty_reduce(v); }
};
static inline F operator+(F a, F b) { uint32_t s = a.v + b.v; return F::raw(s >= P ? s - P : s); }
static inline F operator-(F a, F b) { return F::raw(a.v >= b.v ? a.v - b.v : a.v + P - b.v); }
static inline F operator*(F a, F b) { return F::raw(monty_reduce((uint64_t)a.v * b.v)); }
static inline F& operator+=(F& a, F b) { a = a + b; return a; }
static inline F& operator*=(F& a, F b) { a = a * b; return a; }
static inline F fpow(F b, uint64_t e) { F r = F::one(); while (e) { if (e & 1) r *= b; b *= b; e >>= 1; } return r; }
static inline F two_adic_generator(int bits) {    // 31^((p - 1) / 2^27), squared down
    F g = fpow(F::from_canonical(31), (P - 1) >> TWO_ADICITY);
    for (int i = bits; i < TWO_ADICITY; i++) g *= g;
    return g;
}

// ---- Poseidon2 (x^7, 8 + 13 rounds)
struct P2Constants {
    F ext[8][16], internal[13];
    P2Constants() {
        static const uint32_t rc[30][16] = {
#include "bb_poseidon2_rc.inc"
        };
        for (int r = 0; r < 4; r++)
            for (int i = 0; i < 16; i++) { ext[r][i] = F::from_canonical(rc[r][i]); ext[4 + r][i] = F::from_canonical(rc[17 + r][i]); }
        for (int r = 0; r < 13; r++) internal[r] = F::from_canonical(rc[4 + r][0]);
    }
};
static inline const P2Constants& p2c() { static const P2Constants c; return c; }
static inline void m4(F* x) {
    F t01 = x[0] + x[1], t23 = x[2] + x[3], t0123 = t01 + t23;
    F t01123 = t0123 + x[1], t01233 = t0123 + x[3];
    F n3 = t01233 + (x[0] + x[0]), n1 = t01123 + (x[2] + x[2]), n0 = t01123 + t01, n2 = t01233 + t23;
    x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3;
}
static inline void external_linear(F* s) {
    for (int j = 0; j < 16; j += 4) m4(s + j);
    F sums[4];
    for (int k = 0; k < 4; k++) sums[k] = s[k] + s[k + 4] + s[k + 8] + s[k + 12];
    for (int j = 0; j < 16; j++) s[j] += sums[j & 3];
}
static inline void internal_linear(F* s) {       // (sum + d_i s_i) 2^-32, d = [-2, 1, 2, 4, ..., 2^13, 2^15]   (UNPINNED, see header)
    static const int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
    uint64_t sum = 0;
    for (int i = 0; i < 16; i++) sum += s[i].v;
    uint64_t v0 = s[0].v, neg0 = v0 ? P - v0 : 0;
    uint32_t n0 = monty_reduce(sum - v0 + neg0);
    for (int i = 1; i < 16; i++) s[i].v = monty_reduce(sum + ((uint64_t)s[i].v << SH[i - 1]));
    s[0].v = n0;
}
static inline F sbox(F x) { F x2 = x * x, x3 = x2 * x, x4 = x2 * x2; return x4 * x3; }
static inline void permute(F* s) {
    const P2Constants& c = p2c();
    external_linear(s);
    for (int r = 0; r < 4; r++) { for (int i = 0; i < 16; i++) s[i] = sbox(s[i] + c.ext[r][i]); external_linear(s); }
    for (int r = 0; r < 13; r++) { s[0] = sbox(s[0] + c.internal[r]); internal_linear(s); }
    for (int r = 4; r < 8; r++) { for (int i = 0; i < 16; i++) s[i] = sbox(s[i] + c.ext[r][i]); external_linear(s); }
}
struct Digest { F d[8]; };
struct Sponge {
    F s[16];
    int fill = 0;
    Sponge() { for (auto& x : s) x = F::zero(); }
    void absorb(F x) { s[fill++] = x; if (fill == 8) { permute(s); fill = 0; } }
    Digest finish() { if (fill) { permute(s); fill = 0; } Digest d; for (int i = 0; i < 8; i++) d.d[i] = s[i]; return d; }
};
static inline Digest compress(const Digest& l, const Digest& r) {
    F s[16];
    for (int i = 0; i < 8; i++) { s[i] = l.d[i]; s[8 + i] = r.d[i]; }
    permute(s);
    Digest d;
    for (int i = 0; i < 8; i++) d.d[i] = s[i];
    return d;
}

// in: [n][w] row-major coefficients; out: [n << log_blowup][w]; out[bitrev(k)] = sum_i in[i] w_N^{ki}   (encoder.rs:L22-L38)
static inline void rs_encode(const F* in, int log_n, int w, int log_blowup, F* out) {
    const int log_N = log_n + log_blowup;
    const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
    memcpy(out, in, n * w * sizeof(F));
    memset((void*)(out + n * w), 0, (N - n) * w * sizeof(F));
    if (log_N == 0) return;
    std::vector<F> tw(N / 2);
    { F g = two_adic_generator(log_N), cur = F::one(); for (size_t i = 0; i < N / 2; i++) { tw[i] = cur; cur *= g; } }
    for (int s = log_N; s >= 1; s--) {
        const size_t half = (size_t)1 << (s - 1), stride = N >> s;
#pragma omp parallel for schedule(static)
        for (size_t idx = 0; idx < N / 2; idx++) {
            size_t blk = idx / half, j = idx % half;
            F* a = out + (blk * 2 * half + j) * w;
            F* b = a + half * w;
            F t = tw[j * stride];
            for (int c = 0; c < w; c++) { F x = a[c], y = b[c]; a[c] = x + y; b[c] = (x - y) * t; }
        }
    }
}

// commit_tensors over codewords of height h (p3sync.rs:L40-L143): layers leaf-first back to back, root, commitment
static inline void merkle_commit(const std::vector<const F*>& ts, const std::vector<int>& widths, size_t h,
                                 std::vector<Digest>* tree, Digest* commit) {
    int log_h = 0;
    while (((size_t)1 << log_h) < h) log_h++;
    size_t total_w = 0;
    for (int w : widths) total_w += w;
    tree->assign(2 * h - 1, Digest{});
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < h; i++) {
        Sponge sp;
        for (size_t t = 0; t < ts.size(); t++)
            for (int c = 0; c < widths[t]; c++) sp.absorb(ts[t][i * widths[t] + c]);
        (*tree)[i] = sp.finish();
    }
    size_t off = 0;
    for (size_t n = h; n > 1; n /= 2) {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n / 2; i++) (*tree)[off + n + i] = compress((*tree)[off + 2 * i], (*tree)[off + 2 * i + 1]);
        off += n;
    }
    F meta[2] = {F::from_canonical((uint32_t)log_h), F::from_canonical((uint32_t)total_w)};
    Sponge sp;
    sp.absorb(meta[0]); sp.absorb(meta[1]);
    *commit = compress((*tree)[2 * h - 2], sp.finish());
}

}  // namespace orcbb
