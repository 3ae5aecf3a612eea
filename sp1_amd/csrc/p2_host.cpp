// sp1_amd/csrc/p2_host.cpp — the Fiat–Shamir transcript's Poseidon2 permutation on the HOST (x86-64, no device code in
// this translation unit).
//
// Why it exists: the transcript is a duplex sponge, one permutation per 8 observed words, each depending on the one
// before; a core-shard proof observes ~45k words (the LogUp-GKR circuit output and round messages, the opened values of
// every chip, the jagged and BaseFold messages), i.e. ~6k strictly sequential permutations during which the GPU waits.
// The per-lane formulation of poseidon2.hpp, run as scalar host code, takes ~0.9 us per permutation (throughput-bound:
// ~600 Montgomery products on one core): 5-6 ms of a ~87 ms proof. A single permutation has 16-way data parallelism
// in every layer, which is exactly one AVX-512 register pair of 64-bit lanes:
//   * the state lives in two __m512i of 8 x u64 (words 0..7, 8..15), so vpmuludq multiplies lanes in place (no even/odd
//     shuffling) and the linear layers add WITHOUT modular corrections (64-bit head-room);
//   * external rounds: (x + rc) is brought under 2^31.5 with 2^31 == 2^24 - 1 (mod p), cubed with two Montgomery
//     products, and the 4x4 circulant is out_i = T + (x_i + x_{i+1}) + x_{i+1} with T the block sum (two in-lane
//     rotations), the column sums two more shuffles;
//   * internal rounds: the diagonal on Montgomery words is [-2, 1, 2, 4, .., 2^13, 2^15] * 2^-32 (poseidon2.hpp), so a
//     round is one variable shift, one add of the broadcast sum and one Montgomery reduction for all 16 lanes; the
//     S-box of word 0 runs on the low 128-bit lane while the sum of the other 15 words is reduced next to it.
// Same field elements as p2::permute / p2::permute_int (tests/test_host_permutation.py compares the forms word for word,
// and every proof the GPU tests check goes through this transcript). CPUs without AVX-512 run the scalar integer form.
#include <immintrin.h>
#include <stdint.h>

#include "common.hpp"
#include "poseidon2.hpp"

namespace sp1hip {

#define P2H_TARGET __attribute__((target("avx512f,avx512dq,avx512vl,avx512bw")))

namespace {

struct VecConstants {
    alignas(64) uint64_t ext[8][16];
    alignas(16) uint64_t internal[20][2];
};

const p2::RoundConstants& scalar_rc() {
    static const p2::RoundConstants rc = p2::make_round_constants();
    return rc;
}

const VecConstants& vec_rc() {
    static const VecConstants vc = [] {
        VecConstants v;
        const p2::RoundConstants& rc = scalar_rc();
        for (int r = 0; r < 8; r++)
            for (int i = 0; i < 16; i++) v.ext[r][i] = rc.ext[r][i];
        for (int r = 0; r < 20; r++) { v.internal[r][0] = rc.internal[r]; v.internal[r][1] = 0; }
        return v;
    }();
    return vc;
}

// x < 2^63 per 64-bit lane -> (x + t p) >> 32 == x 2^-32 (mod p), in [x / 2^32, x / 2^32 + p)
P2H_TARGET inline __m512i mred(__m512i x, __m512i nmu, __m512i p) {
    const __m512i t = _mm512_mul_epu32(x, nmu);               // low word = lo32(x) * (-p^-1) mod 2^32
    return _mm512_srli_epi64(_mm512_add_epi64(_mm512_mul_epu32(t, p), x), 32);
}
P2H_TARGET inline __m128i mred128(__m128i x, __m128i nmu, __m128i p) {
    const __m128i t = _mm_mul_epu32(x, nmu);
    return _mm_srli_epi64(_mm_add_epi64(_mm_mul_epu32(t, p), x), 32);
}
// [0, 2p) -> [0, p); more generally r < 2^32 -> min(r, r - p mod 2^32) (the high words are zero and stay zero)
P2H_TARGET inline __m512i correct(__m512i r, __m512i p) { return _mm512_min_epu32(r, _mm512_sub_epi32(r, p)); }
P2H_TARGET inline __m128i correct128(__m128i r, __m128i p) { return _mm_min_epu32(r, _mm_sub_epi32(r, p)); }

// v < 53 * 2^31 -> a word < 2^31.5 congruent to it: 2^31 == 2^24 - 1 (mod p)
P2H_TARGET inline __m512i fold31(__m512i v) {
    const __m512i hi = _mm512_srli_epi64(v, 31);
    const __m512i lo = _mm512_and_si512(v, _mm512_set1_epi64(0x7fffffff));
    return _mm512_sub_epi64(_mm512_add_epi64(lo, _mm512_slli_epi64(hi, 24)), hi);
}

// (x + rc)^3 R^-2, canonical. x <= 36 * 2^31 - rc.
P2H_TARGET inline __m512i sbox_ext(__m512i x, __m512i rc, __m512i nmu, __m512i p) {
    const __m512i y = fold31(_mm512_add_epi64(x, rc));                          // < 2^31 + 36 * 2^24 < 2^31.5
    const __m512i y2 = correct(mred(_mm512_mul_epu32(y, y), nmu, p), p);        // y^2 < 2^63; < p
    return correct(mred(_mm512_mul_epu32(y2, y), nmu, p), p);                   // < p
}

// inputs < 2^32 (canonical: < p) -> outputs <= 35 max(input), no reductions
P2H_TARGET inline void external_linear(__m512i& a, __m512i& b) {
    {
        const __m512i r1 = _mm512_permutex_epi64(a, 0x39), u = _mm512_add_epi64(a, r1);
        const __m512i t = _mm512_add_epi64(u, _mm512_permutex_epi64(u, 0x4E));
        a = _mm512_add_epi64(_mm512_add_epi64(t, u), r1);
    }
    {
        const __m512i r1 = _mm512_permutex_epi64(b, 0x39), u = _mm512_add_epi64(b, r1);
        const __m512i t = _mm512_add_epi64(u, _mm512_permutex_epi64(u, 0x4E));
        b = _mm512_add_epi64(_mm512_add_epi64(t, u), r1);
    }
    const __m512i s = _mm512_add_epi64(a, b);
    const __m512i s2 = _mm512_add_epi64(s, _mm512_shuffle_i64x2(s, s, 0x4E));
    a = _mm512_add_epi64(a, s2);
    b = _mm512_add_epi64(b, s2);
}

P2H_TARGET void permute_avx512(uint32_t* state) {
    const VecConstants& vc = vec_rc();
    const __m512i p = _mm512_set1_epi64(kb::P), nmu = _mm512_set1_epi64(kb::NMU);
    const __m128i p1 = _mm_set1_epi64x(kb::P), nmu1 = _mm_set1_epi64x(kb::NMU);
    __m512i a = _mm512_cvtepu32_epi64(_mm256_loadu_si256((const __m256i*)state));
    __m512i b = _mm512_cvtepu32_epi64(_mm256_loadu_si256((const __m256i*)(state + 8)));
    external_linear(a, b);
    for (int r = 0; r < 4; r++) {
        a = sbox_ext(a, _mm512_load_si512(&vc.ext[r][0]), nmu, p);
        b = sbox_ext(b, _mm512_load_si512(&vc.ext[r][8]), nmu, p);
        external_linear(a, b);
    }
    // internal rounds: every word lazy in [0, p + 2^16)
    a = correct(fold31(a), p);
    b = correct(fold31(b), p);
    const __m512i ka = _mm512_setr_epi64(1, 0, 1, 2, 3, 4, 5, 6), kb_ = _mm512_setr_epi64(7, 8, 9, 10, 11, 12, 13, 15);
    // Word 0 lives in its own 128-bit register through the internal rounds: the chain S-box -> new word 0 -> next S-box is the
    // critical path of the permutation, and it does not have to wait for the broadcast + vector reduction the other 15 words
    // go through (new_0 = (rest + 2p - t) 2^-32 needs only `rest`, which is ready long before t).
    __m128i s0 = _mm512_castsi512_si128(a);
    const __m128i two_p1 = _mm_set1_epi64x(2 * (uint64_t)kb::P);
    for (int r = 0; r < 20; r++) {
        // word 0 through the S-box: t = (s0 + rc)^3 R^-2 in [0, 2p)
        const __m128i y = correct128(_mm_add_epi64(s0, _mm_load_si128((const __m128i*)vc.internal[r])), p1);   // < p + 2^16
        const __m128i y2 = mred128(_mm_mul_epu32(y, y), nmu1, p1);              // < y^2 / 2^32 + p < 1.5 p: y2 y < 2^63 as it is
        const __m128i t = mred128(_mm_mul_epu32(y2, y), nmu1, p1);              // < 0.75 p + p
        // meanwhile: the sum of words 1..15 in every lane
        __m512i rest = _mm512_add_epi64(_mm512_maskz_mov_epi64(0xFE, a), b);
        rest = _mm512_add_epi64(rest, _mm512_shuffle_i64x2(rest, rest, 0x4E));
        rest = _mm512_add_epi64(rest, _mm512_shuffle_i64x2(rest, rest, 0xB1));
        rest = _mm512_add_epi64(rest, _mm512_shuffle_epi32(rest, (_MM_PERM_ENUM)0x4E));
        // new word 0 = (sum - 2 t) 2^-32 = (rest + 2p - t) 2^-32 with sum = rest + t: < p + 2^5
        s0 = mred128(_mm_sub_epi64(_mm_add_epi64(_mm512_castsi512_si128(rest), two_p1), t), nmu1, p1);
        // new_i = (sum + 2^k_i s_i) 2^-32 for the other words (lane 0 of `a` is dead weight until the rounds are over)
        const __m512i sum = _mm512_add_epi64(rest, _mm512_broadcastq_epi64(t));
        a = mred(_mm512_add_epi64(_mm512_sllv_epi64(a, ka), sum), nmu, p);
        b = mred(_mm512_add_epi64(_mm512_sllv_epi64(b, kb_), sum), nmu, p);
    }
    a = _mm512_mask_mov_epi64(a, 1, _mm512_castsi128_si512(s0));
    for (int r = 4; r < 8; r++) {
        a = sbox_ext(a, _mm512_load_si512(&vc.ext[r][0]), nmu, p);
        b = sbox_ext(b, _mm512_load_si512(&vc.ext[r][8]), nmu, p);
        external_linear(a, b);
    }
    a = correct(fold31(a), p);      // <= 35 p -> < 2^31.5 < 2p -> canonical
    b = correct(fold31(b), p);
    _mm256_storeu_si256((__m256i*)state, _mm512_cvtepi64_epi32(a));
    _mm256_storeu_si256((__m256i*)(state + 8), _mm512_cvtepi64_epi32(b));
}

bool cpu_has_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
                           __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw");
    return ok;
}

}  // namespace

void p2_host_permute(uint32_t (&state)[16]) {
    if (cpu_has_avx512()) permute_avx512(state);
    else p2::permute_int(state, scalar_rc());
}

}  // namespace sp1hip

extern "C" {

int sp1hip_poseidon2_permute_host(uint32_t* h_states, size_t n, int form) {
    SP1HIP_REQUIRE(h_states || n == 0, "null states");
    SP1HIP_REQUIRE(form >= 0 && form <= 2, "form: 0 = the transcript's permutation, 1 = scalar integer form, 2 = scalar fp64 form");
    for (size_t i = 0; i < n; i++) {
        uint32_t (&s)[16] = *reinterpret_cast<uint32_t (*)[16]>(h_states + 16 * i);
        for (int k = 0; k < 16; k++) SP1HIP_REQUIRE(s[k] < kb::P, "state words must be canonical Montgomery words");
        if (form == 0) sp1hip::p2_host_permute(s);
        else if (form == 1) p2::permute_int(s, sp1hip::scalar_rc());
        else p2::permute(s, sp1hip::scalar_rc());
    }
    return SP1HIP_SUCCESS;
}

int sp1hip_host_permutation_is_vectorised(void) { return sp1hip::cpu_has_avx512() ? 1 : 0; }

}  // extern "C"
