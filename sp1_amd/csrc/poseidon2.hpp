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
    double ext_magic[8][16];   // MAGIC + (ext[][] as centred representative in [-(p-1)/2, (p-1)/2])
    uint32_t internal_neg[20]; // internal[r] - p as a two's-complement word (in [-p, 0))
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
    for (int r = 0; r < 8; r++)
        for (int i = 0; i < 16; i++) {
            const int32_t c = rc.ext[r][i] > (kb::P - 1) / 2 ? (int32_t)(rc.ext[r][i] - kb::P) : (int32_t)rc.ext[r][i];
            rc.ext_magic[r][i] = 6755399441055744.0 + (double)c;
        }
    for (int r = 0; r < 20; r++) rc.internal_neg[r] = rc.internal[r] - kb::P;
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

// Wave-uniform multiplier 2^k that the optimiser cannot see through (so `s * m + sum` stays one
// v_mad_u64_u32 instead of a 64-bit shift plus a 64-bit add). Host: plain value.
KB_HD uint32_t opaque_pow2(int k) {
    uint32_t m = 1u << k;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(m));
#endif
    return m;
}

// Internal rounds keep ALL lanes LAZY: unsigned words congruent to the true value, lanes 1..15 in
// [0, p + 2^15), lane 0 in [0, p + 8). new_i = (sum + 2^k s_i) * 2^-32 is one multiply-add
// (V = s_i * 2^k + sum < 2^48), one v_mul_lo_u32 and one v_mad_u64_u32 (additive Montgomery form,
// result in [V/2^32, V/2^32 + p), i.e. again < p + 2^15): 3 VALU instructions per lane per round, no
// correction. The diagonal is [-2, 1, 2, 4, .., 2^13, 2^15] on Montgomery words and the division by
// 2^32 is the reference's MONTY_INVERSE factor
// (/root/reference/sp1-gpu/crates/sys/include/poseidon2/poseidon2_kb31_16.cuh:L118-L140).
// Lane 0 goes through the S-box in SIGNED form (see monty_reduce_signed below): y = s_0 + (rc - p)
// is in [-p, p + 8), its cube comes back as o in (-p, p), and t = o + p in (0, 2p) is the lazy
// unsigned word that enters the sum; V_0 = sum + 2 (2p - t) >= 0 is == sum - 2 t (mod p) and
// V_0 < 2^35, so new_0 < p + 8. No conditional correction in the whole round.
KB_HD int32_t monty_reduce_signed(int64_t x);

// acc + a as ONE v_mad_u64_u32 (a * 1 + acc): a 64-bit add of a 32-bit word without building its
// zero-extended register pair first
KB_HD uint64_t acc_u32(uint32_t a, uint64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, 1, %2" : "=v"(r) : "v"(a), "v"(acc) : "vcc");
    return r;
#else
    return acc + a;
#endif
}

KB_HD void internal_round_lazy(uint32_t (&s)[16], uint32_t rc_minus_p) {
    const int32_t y = (int32_t)(s[0] + rc_minus_p);
    const int32_t y2 = monty_reduce_signed((int64_t)y * y);
    const uint32_t t = (uint32_t)monty_reduce_signed((int64_t)y2 * y) + kb::P;
    // pair sums of lanes 1..15 fit 32 bits (2 (p + 2^15) < 2^32; t < 2p must stay alone); the 64-bit
    // accumulation is a chain of multiply-adds by one (no zero-extension moves)
    uint64_t sum = acc_u32(s[15], acc_u32(t, 0));
    sum = acc_u32(s[1] + s[2], sum);
    sum = acc_u32(s[3] + s[4], sum);
    sum = acc_u32(s[5] + s[6], sum);
    sum = acc_u32(s[7] + s[8], sum);
    sum = acc_u32(s[9] + s[10], sum);
    sum = acc_u32(s[11] + s[12], sum);
    sum = acc_u32(s[13] + s[14], sum);
    const uint64_t v0 = (uint64_t)(2 * kb::P - t) * opaque_pow2(1) + sum;
    constexpr int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
#pragma unroll
    for (int i = 1; i < 16; i++) s[i] = kb::monty_reduce_lazy((uint64_t)s[i] * opaque_pow2(SH[i - 1]) + sum);
    s[0] = kb::monty_reduce_lazy(v0);
}

// all-integer, canonical-lane-0 form of the internal linear layer (used by permute_int)
KB_HD void internal_linear_lazy(uint32_t (&s)[16]) {
    uint64_t sum = (uint64_t)(s[0] + s[1]) + (s[2] + s[3]) + (s[4] + s[5]) + (s[6] + s[7]) + (s[8] + s[9]) +
                   (s[10] + s[11]) + (s[12] + s[13]) + (s[14] + s[15]);
    const uint64_t v0 = (uint64_t)(kb::P - s[0]) * opaque_pow2(1) + sum;
    constexpr int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
#pragma unroll
    for (int i = 1; i < 16; i++) s[i] = kb::monty_reduce_lazy((uint64_t)s[i] * opaque_pow2(SH[i - 1]) + sum);
    s[0] = kb::monty_reduce(v0);
}

// (s + rc)^3: the square is left in [0, 2p) (2p * p < 2^32 p keeps the next reduction valid).
KB_HD uint32_t sbox(uint32_t s, uint32_t rc) {
    const uint32_t x = kb::add(s, rc);
    const uint32_t x2 = kb::monty_reduce_lazy((uint64_t)x * x);
    return kb::monty_reduce((uint64_t)x2 * x);
}

// ---- external rounds: exact fp64 linear layer + signed, correction-free S-box --------------------
// gfx950 issues v_add_f64 / v_mul_f64 / v_fma_f64 / v_rndne_f64 / v_cvt_* at the rate of any other
// VOP3 integer op (profiles/r01_ubench_f64.txt), and a double holds integers up to 2^53 exactly. So
// the external linear layer is done on doubles WITHOUT any modular reduction: 72 v_add_f64 instead of
// 72 modular adds (3 VALU each). Inputs are integers |x| < p, every output is a combination with
// coefficient sum <= 35, i.e. |out| < 2^37: exact.
// The S-box takes such an unreduced integer v (a Montgomery word up to a multiple of p):
//   q = rndne(v / p)                 (v_mul_f64, v_rndne_f64; the estimate is off by < 2^-41 and p is
//                                     odd, so q is the exact nearest integer and |v - q p| <= (p-1)/2)
//   t = fma(-q, p, v + (MAGIC + rc_centred))   MAGIC = 1.5 * 2^52: t is exact, its ulp is 1, and the low
//                                     32 bits of its mantissa ARE the two's-complement int32
//   y = lo32(t) = (v - q p) + rc_centred       (|y| <= p - 1; no conversion instruction, no integer add)
//   s = (y*y + q p) >> 32            q = int32(lo32(y*y) * -p^-1): SIGNED quotient digit, so that
//   o = (s*y + q' p) >> 32           |s|, |o| < p/2 + p^2/2^32 < p  (v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32)
// and hands back double(o): a signed representative of (v + rc)^3 R^-2, no conditional correction
// anywhere. Per lane 11 VALU instructions, per external round 16*11 + 64 = 240 against 400 for the
// all-integer form (the compiler turns the x + x terms of the M4 blocks into v_fma_f64); the results are the same field elements (the GPU tests compare every digest with
// the oracle bit for bit).
constexpr double P_F64 = 2130706433.0;
constexpr double INV_P_F64 = 1.0 / 2130706433.0;
constexpr double MAGIC_F64 = 6755399441055744.0;   // 1.5 * 2^52

KB_HD uint32_t lo32_of(double t) { return (uint32_t)__builtin_bit_cast(uint64_t, t); }

KB_HD void m4_f64(double& x0, double& x1, double& x2, double& x3) {
    double t01 = x0 + x1, t23 = x2 + x3;
    double t0123 = t01 + t23;
    double t01123 = t0123 + x1, t01233 = t0123 + x3;
    double n3 = t01233 + (x0 + x0);
    double n1 = t01123 + (x2 + x2);
    double n0 = t01123 + t01;
    double n2 = t01233 + t23;
    x0 = n0; x1 = n1; x2 = n2; x3 = n3;
}

KB_HD void external_linear_f64(double (&d)[16]) {
#pragma unroll
    for (int j = 0; j < 16; j += 4) m4_f64(d[j], d[j + 1], d[j + 2], d[j + 3]);
    double sums[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sums[k] = (d[k] + d[k + 4]) + (d[k + 8] + d[k + 12]);
#pragma unroll
    for (int j = 0; j < 16; j++) d[j] = d[j] + sums[j & 3];
}

// signed Montgomery reduction: (x + q p) / 2^32 with q in [-2^31, 2^31); |x| < 2^62 -> |result| < |x|/2^32 + p/2
KB_HD int32_t monty_reduce_signed(int64_t x) {
    const int32_t q = (int32_t)((uint32_t)x * kb::NMU);
    return (int32_t)((x + (int64_t)q * (int32_t)kb::P) >> 32);
}

KB_HD double sbox_f64(double v, double magic_plus_rc) {
    const double q = __builtin_rint(v * INV_P_F64);
    const int32_t y = (int32_t)lo32_of(__builtin_fma(-q, P_F64, v + magic_plus_rc));
    const int32_t y2 = monty_reduce_signed((int64_t)y * y);
    return (double)monty_reduce_signed((int64_t)y2 * y);
}

// exact integer v, |v| < 2^44  ->  the word in [0, p] congruent to it (p itself only when p | v)
KB_HD uint32_t reduce_f64(double v) {
    const double q = __builtin_floor(v * INV_P_F64);
    return lo32_of(__builtin_fma(-q, P_F64, v + MAGIC_F64));
}

// The permutation on a state held as doubles. In: exact integers |d_i| < 2^38 congruent to the state words (fresh
// Montgomery words, or the UNREDUCED outputs of a previous call: the first linear layer then yields < 35 * 2^38 <
// 2^44, which the S-box's reduction takes). Out: exact integers |d_i| < 35 p < 2^37, unreduced. A sponge that
// absorbs block after block keeps its capacity lanes in this form and never pays for reducing / re-converting them,
// and the rate lanes it is about to overwrite are never reduced at all (permute() below reduces all 16).
template <class RC>
KB_HD void permute_f64(double (&d)[16], const RC& rc) {
    external_linear_f64(d);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = sbox_f64(d[i], rc.ext_magic[r][i]);
        external_linear_f64(d);
    }
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = reduce_f64(d[i]);            // [0, p]: inside the lazy ranges
#pragma unroll 2
    for (int r = 0; r < 20; r++) internal_round_lazy(s, rc.internal_neg[r]);
    // all lanes < p + 2^15: any representative works for the exact layer
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = (double)s[i];
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = sbox_f64(d[i], rc.ext_magic[r][i]);
        external_linear_f64(d);
    }
}

// unreduced output lane of permute_f64 -> canonical word
KB_HD uint32_t canonical_f64(double v) {
    const uint32_t w = reduce_f64(v);
    return kb::umin(w, w - kb::P);
}

template <class RC>
KB_HD void permute(uint32_t (&s)[16], const RC& rc) {
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = (double)s[i];
    permute_f64(d, rc);
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = canonical_f64(d[i]);
}

// all-integer form of the same permutation (the previous production path; kept for the A/B
// micro-benchmark and as an in-library cross-check of the fp64 formulation)
template <class RC>
KB_HD void permute_int(uint32_t (&s)[16], const RC& rc) {
    external_linear(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox(s[i], rc.ext[r][i]);
        external_linear(s);
    }
#pragma unroll 1
    for (int r = 0; r < 20; r++) {
        s[0] = sbox(s[0], rc.internal[r]);
        internal_linear_lazy(s);
    }
#pragma unroll
    for (int i = 1; i < 16; i++) s[i] = kb::umin(s[i], s[i] - kb::P);   // lazy -> canonical
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox(s[i], rc.ext[r][i]);
        external_linear(s);
    }
}

// ---- cooperative form: ONE permutation per 16 lanes (a DPP row), lane r holds state word r --------------------------
// For the latency-bound places (the top of a Merkle tree: a chain of dependent compressions with fewer nodes than
// lanes). The per-lane form above issues ~3.7k dependent-ish VALU instructions per permutation from one wave; here
// the linear layers become wavefront shuffles — quad_perm for the 4x4 MDS blocks, row_ror for the column sums and
// the internal layer's 16-term sum — and a permutation is ~1.3k instructions per wave, i.e. ~2.8x less latency.
// All lanes of the row must be active. Words are canonical ([0, p)) throughout.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
#else
    return v;       // device-only; the host pass just needs the declaration
#endif
}
constexpr int DPP_QUAD_SWAP1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_SWAP2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int DPP_QUAD_NEXT = 0x39;    // quad_perm [1,2,3,0]: lane i reads lane i+1 of its quad
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

// y_i = 2 x_i + 3 x_{i+1} + x_{i+2} + x_{i+3} inside each quad (the circulant the m4() sequence computes), then every
// word gets the sum of the four quads' words at its position.
__device__ __forceinline__ uint32_t external_linear_coop(uint32_t x) {
    const uint32_t a = kb::add(x, dpp_mov<DPP_QUAD_SWAP1>(x));
    const uint32_t quad_sum = kb::add(a, dpp_mov<DPP_QUAD_SWAP2>(a));
    const uint32_t x1 = dpp_mov<DPP_QUAD_NEXT>(x);
    const uint32_t y = kb::add(kb::add(quad_sum, x), kb::add(x1, x1));
    const uint32_t u = kb::add(y, dpp_mov<DPP_ROW_ROR8>(y));
    const uint32_t v = kb::add(u, dpp_mov<DPP_ROW_ROR4>(u));
    return kb::add(y, v);
}

// lane = index of this lane in its row (0..15)
template <class RC>
__device__ __forceinline__ uint32_t permute_coop16(uint32_t x, uint32_t lane, const RC& rc) {
    // internal diagonal on Montgomery words: [-2, 1, 2, 4, ..., 2^13, 2^15] (lane 0 enters as p - s_0 with shift 1)
    const uint32_t shift = lane == 0 ? 1u : (lane == 15 ? 15u : lane - 1);
    x = external_linear_coop(x);
#pragma unroll 1
    for (int r = 0; r < 4; r++) x = external_linear_coop(sbox(x, rc.ext[r][lane]));
#pragma unroll 1
    for (int r = 0; r < 20; r++) {
        const uint32_t cubed = sbox(x, rc.internal[r]);
        x = lane == 0 ? cubed : x;
        uint32_t t = kb::add(x, dpp_mov<DPP_ROW_ROR8>(x));
        t = kb::add(t, dpp_mov<DPP_ROW_ROR4>(t));
        t = kb::add(t, dpp_mov<DPP_ROW_ROR2>(t));
        t = kb::add(t, dpp_mov<DPP_ROW_ROR1>(t));            // sum of the 16 words mod p, in every lane
        const uint32_t in = lane == 0 ? kb::P - x : x;
        x = kb::monty_reduce(((uint64_t)in << shift) + t);
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) x = external_linear_coop(sbox(x, rc.ext[r][lane]));
    return x;
}

}  // namespace p2

// The transcript's permutation on the host (p2_host.cpp): AVX-512 where the CPU has it, the scalar integer form otherwise.
namespace sp1hip {
void p2_host_permute(uint32_t (&state)[16]);
}
