// oracle/kb_hash.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// Poseidon2-KoalaBear width 16 (8 full + 20 partial rounds, x^3), the padding-free overwrite
// sponge, the truncated-permutation 2-to-1 compressor and the duplex challenger.
//
//   permutation type + parameters   /root/reference/slop/crates/koala-bear/src/koala_bear_poseidon2.rs:L20-L63
//   round structure (restated in-tree) /root/reference/crates/hypercube/src/operations/poseidon2/trace.rs:L29-L152
//   M4 / external / internal layers /root/reference/crates/hypercube/src/operations/poseidon2/air.rs:L17-L66
//   internal diagonal [-2,1,2,..,2^13,2^15] applied to Montgomery words + one reduce (the 2^-32)
//                                   /root/reference/sp1-gpu/crates/sys/include/poseidon2/poseidon2_kb31_16.cuh:L118-L140
//   PaddingFreeSponge<16,8,8>, TruncatedPermutation<2,8,16>  koala_bear_poseidon2.rs:L33-L41
//   DuplexChallenger<F,Perm,16,8>   /root/reference/slop/crates/challenger/src/lib.rs:L25-L87 (type),
//                                   semantics restated in /root/reference/sp1-gpu/crates/sys/include/challenger/challenger.cuh:L13-L118
// The permutation/sponge/challenger code itself is in un-vendored Plonky3 (p3-poseidon2,
// p3-symmetric, p3-challenger =0.4.3-succinct). Permutation, sponge and compressor are pinned by
// tests/golden (real proof data) and the SURVEY App. A known answers; the challenger is pinned by
// replaying the reference's real shard-proof transcript (tests/golden/make_transcript.py: grinding
// witnesses, sumcheck points, fold betas and query indices all reproduce).
#pragma once
#include <vector>

#include "kb_field.hpp"

namespace orc {

constexpr int P2_WIDTH = 16, P2_RATE = 8, P2_DIGEST = 8;

struct P2Constants {
    F ext[8][16];
    F internal[20];
    P2Constants() {
        static const uint32_t rc[28][16] = {
#include "kb_poseidon2_rc.inc"
        };
        for (int r = 0; r < 4; r++)
            for (int i = 0; i < 16; i++) {
                ext[r][i] = F::from_canonical(rc[r][i]);
                ext[4 + r][i] = F::from_canonical(rc[24 + r][i]);
            }
        for (int r = 0; r < 20; r++) internal[r] = F::from_canonical(rc[4 + r][0]);
    }
};
static inline const P2Constants& p2c() { static const P2Constants c; return c; }

static inline void m4(F* x) {
    F t01 = x[0] + x[1], t23 = x[2] + x[3];
    F t0123 = t01 + t23;
    F t01123 = t0123 + x[1], t01233 = t0123 + x[3];
    F n3 = t01233 + (x[0] + x[0]);
    F n1 = t01123 + (x[2] + x[2]);
    F n0 = t01123 + t01;
    F n2 = t01233 + t23;
    x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3;
}

static inline void external_linear(F* s) {
    for (int j = 0; j < 16; j += 4) m4(s + j);
    F sums[4];
    for (int k = 0; k < 4; k++) sums[k] = s[k] + s[k + 4] + s[k + 8] + s[k + 12];
    for (int j = 0; j < 16; j++) s[j] += sums[j & 3];
}

static inline void internal_linear(F* s) {
    // new_i = (sum + d_i * s_i) * 2^-32, d = [-2, 1, 2, 4, ..., 2^13, 2^15], on Montgomery words
    static const int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
    uint64_t sum = 0;
    for (int i = 0; i < 16; i++) sum += s[i].v;
    uint64_t v0 = s[0].v, neg0 = v0 ? KB_P - v0 : 0;
    uint32_t n0 = monty_reduce(sum - v0 + neg0);
    for (int i = 1; i < 16; i++) s[i].v = monty_reduce(sum + ((uint64_t)s[i].v << SH[i - 1]));
    s[0].v = n0;
}

static inline F cube(F x) { return x * x * x; }

static inline void permute(F* s) {
    const P2Constants& c = p2c();
    external_linear(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = cube(s[i] + c.ext[r][i]);
        external_linear(s);
    }
    for (int r = 0; r < 20; r++) {
        s[0] = cube(s[0] + c.internal[r]);
        internal_linear(s);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 16; i++) s[i] = cube(s[i] + c.ext[r][i]);
        external_linear(s);
    }
}

struct Digest {
    F d[8];
    bool operator==(const Digest& o) const { return !memcmp(d, o.d, sizeof d); }
    bool operator!=(const Digest& o) const { return !(*this == o); }
};

// Overwrite-mode sponge: absorb 8 at a time into state[0..8], permute after every (possibly partial,
// non-empty) block; digest = state[0..8].
struct Sponge {
    F s[16];
    int fill = 0;
    Sponge() { for (auto& x : s) x = F::zero(); }
    void absorb(F x) {
        s[fill++] = x;
        if (fill == P2_RATE) { permute(s); fill = 0; }
    }
    Digest finish() {
        if (fill) { permute(s); fill = 0; }
        Digest d;
        for (int i = 0; i < 8; i++) d.d[i] = s[i];
        return d;
    }
};

static inline Digest hash_slice(const F* xs, size_t n) {
    Sponge sp;
    for (size_t i = 0; i < n; i++) sp.absorb(xs[i]);
    return sp.finish();
}

static inline Digest compress(const Digest& l, const Digest& r) {
    F s[16];
    for (int i = 0; i < 8; i++) { s[i] = l.d[i]; s[8 + i] = r.d[i]; }
    permute(s);
    Digest d;
    for (int i = 0; i < 8; i++) d.d[i] = s[i];
    return d;
}

// ---- Duplex challenger -------------------------------------------------------------------------
struct Challenger {
    F state[16];
    std::vector<F> in, out;
    Challenger() { for (auto& x : state) x = F::zero(); }
    void duplexing() {
        for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
        in.clear();
        permute(state);
        out.assign(state, state + P2_RATE);
    }
    void observe(F x) {
        out.clear();
        in.push_back(x);
        if ((int)in.size() == P2_RATE) duplexing();
    }
    void observe_digest(const Digest& d) { for (int i = 0; i < 8; i++) observe(d.d[i]); }
    void observe_ext(const E& e) { for (int i = 0; i < 4; i++) observe(e.c[i]); }
    F sample() {
        if (!in.empty() || out.empty()) duplexing();
        F r = out.back();
        out.pop_back();
        return r;
    }
    E sample_ext() { E e; for (int i = 0; i < 4; i++) e.c[i] = sample(); return e; }
    uint32_t sample_bits(int bits) { return sample().canonical() & ((1u << bits) - 1); }
    bool check_witness(int bits, F w) { observe(w); return sample_bits(bits) == 0; }
    // Deterministic policy: the SMALLEST canonical witness (the reference uses a parallel
    // find_any, i.e. any valid witness — SURVEY §7 "Transcript determinism").
    F grind(int bits) {
        for (uint32_t i = 0; i < KB_P; i++) {
            Challenger c = *this;
            F w = F::from_canonical(i);
            if (c.check_witness(bits, w)) {
                bool ok = check_witness(bits, w);
                (void)ok;
                return w;
            }
        }
        return F::zero();
    }
};

}  // namespace orc
