// sp1_amd/csrc/poseidon2.hpp — Poseidon2-KoalaBear width 16 (8 full + 20 partial rounds, x^3) for
// gfx950 kernels and for the host-side transcript.
//
// Replaces, for this path, the reference's `KoalaPerm` / `PaddingFreeSponge<_,16,8,8>` /
// `TruncatedPermutation<_,2,8,16>` (/root/reference/slop/crates/koala-bear/src/koala_bear_poseidon2.rs:L20-L63)
// as used by `FieldMerkleTreeProver::commit_tensors`
// (/root/reference/slop/crates/merkle-tree/src/p3sync.rs:L40-L143).
//
// gfx950 shape: one permutation per lane, the 16-word state lives in VGPRs for the whole
// permutation (no LDS, no cross-lane traffic: a leaf absorbs up to 32 blocks back to back, so the
// state never leaves registers between blocks). Round loops are NOT unrolled across rounds (an
// unrolled permutation is ~12k instructions, larger than the instruction cache); round constants
// are wave-uniform and come through the scalar cache (s_load) from __constant__ memory.
#pragma once
#include "kb31.hpp"

namespace p2 {

constexpr int WIDTH = 16, RATE = 8, DIGEST = 8;

struct RoundConstants {
    uint32_t ext[8][16];   // Montgomery form
    uint32_t internal[20];
};

// canonical table generated from the reference (oracle/gen_constants.py)
static const uint32_t RC_CANONICAL[28][16] = {
#include "kb_poseidon2_rc.inc"
};

inline RoundConstants make_round_constants() {
    RoundConstants rc;
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < 16; i++) {
            rc.ext[r][i] = kb::to_monty(RC_CANONICAL[r][i]);
            rc.ext[4 + r][i] = kb::to_monty(RC_CANONICAL[24 + r][i]);
        }
    for (int r = 0; r < 20; r++) rc.internal[r] = kb::to_monty(RC_CANONICAL[4 + r][0]);
    return rc;
}

KB_HD void m4(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
    uint32_t t01 = kb::add(x0, x1), t23 = kb::add(x2, x3);
    uint32_t t0123 = kb::add(t01, t23);
    uint32_t t01123 = kb::add(t0123, x1), t01233 = kb::add(t0123, x3);
    uint32_t n3 = kb::add(t01233, kb::dbl(x0));
    uint32_t n1 = kb::add(t01123, kb::dbl(x2));
    uint32_t n0 = kb::add(t01123, t01);
    uint32_t n2 = kb::add(t01233, t23);
    x0 = n0; x1 = n1; x2 = n2; x3 = n3;
}

KB_HD void external_linear(uint32_t (&s)[16]) {
#pragma unroll
    for (int j = 0; j < 16; j += 4) m4(s[j], s[j + 1], s[j + 2], s[j + 3]);
    uint32_t sums[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sums[k] = kb::add(kb::add(s[k], s[k + 4]), kb::add(s[k + 8], s[k + 12]));
#pragma unroll
    for (int j = 0; j < 16; j++) s[j] = kb::add(s[j], sums[j & 3]);
}

// new_i = (sum + d_i s_i) * 2^-32 with d = [-2, 1, 2, 4, .., 2^13, 2^15] on Montgomery words: one
// 64-bit multiply-add builds sum + (s_i << k), one Montgomery reduction divides by 2^32.
KB_HD void internal_linear(uint32_t (&s)[16]) {
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += s[i];
    // lane 0: sum - s0 + (p - s0) ... with s0 == 0 the reference adds 0, and (sum - 0 + p) reduces to
    // the same residue, so no special case is needed: both are == sum - 2 s0 (mod p) and < 2^32 p.
    uint64_t v0 = sum + kb::P - 2 * (uint64_t)s[0] + kb::P;
    uint32_t n0 = kb::monty_reduce(v0);
    constexpr int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
#pragma unroll
    for (int i = 1; i < 16; i++) s[i] = kb::monty_reduce(sum + ((uint64_t)s[i] << SH[i - 1]));
    s[0] = n0;
}

KB_HD uint32_t cube(uint32_t x) { return kb::mul(kb::sqr(x), x); }

template <class RC>
KB_HD void permute(uint32_t (&s)[16], const RC& rc) {
    external_linear(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = cube(kb::add(s[i], rc.ext[r][i]));
        external_linear(s);
    }
#pragma unroll 1
    for (int r = 0; r < 20; r++) {
        s[0] = cube(kb::add(s[0], rc.internal[r]));
        internal_linear(s);
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = cube(kb::add(s[i], rc.ext[r][i]));
        external_linear(s);
    }
}

}  // namespace p2
