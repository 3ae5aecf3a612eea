// sp1_amd/csrc/zc_keccak.hpp — one Keccak-f round as a fused sub-AIR of the zerocheck (the wide-chip regime, round 5).
//
// KeccakPermute (/root/reference/crates/core/machine/src/syscall/precompiles/keccak256/air.rs:L29-L200) is 2,640 columns and 2,859
// constraints per row — bits of one round's theta / rho / pi / chi / iota — and fills 82 % of a Keccak precompile shard. As
// interpreted bytecode that is 47,768 instruction words per row pair and node in 441 chunks with a 50-register file (three waves
// per CU): 114 ms of round kernels per shard proof against 16 ms for a whole core shard (profiles/r05_precompile_before.txt).
// The constraints are a handful of REGULAR loop nests over (x, y, z), so a caller's program may carry `[16, 5, base_col]` in front of
// the 2,858 asserts that follow `assert_bool(is_real)`; the planner checks the hint against the SSA on a pseudo-random row
// (zerocheck.hip) and the asserts are then evaluated by eleven self-contained pieces with a dozen live values each:
//   q = 0          step flags (24 booleans, their sum, the round index), the bits of A''[0, 0] and the round-constant xor;
//                  also carries the GKR term of the two column groups no constraint reads (export, preimage)
//   q = 1 + x      lane column x: C'[x, z] = xor3(C[x, z], C[x - 1, z], C[x + 1, z - 1]), the parity check of A'[., x, z] against
//                  C'[x, z], and the limbs of A[y, x] from xor3(A'[y, x, z], C[x, z], C'[x, z]) for the five y — C[x], C'[x] and
//                  A'[., x] are loaded once for all of them
//   q = 6 + y      the limbs of A''[y, .]: the five lanes B[., y] = rho-pi(A') bit by bit, chi for the five x at once
// 4,900 column loads per row pair and node (the first cut — one piece per constraint family — had 12,900: the pieces are bound
// by cache bandwidth, not by arithmetic: profiles/r05_precompile_kernel_stats_keccak_pieces_v1.csv)
// Columns: KeccakCols of p3-keccak-air (field order: sp1_amd/machines/riscv_more.py) followed by KeccakMemCols' own seven.
#pragma once
#include "kb31.hpp"
#include "zc_poseidon2.hpp"

namespace sp1hip {

constexpr uint32_t ZC_HINT_KECCAK = 5;
constexpr uint32_t ZC_KK_CONSTRAINTS = 2858, ZC_KK_COLUMNS = 2633, ZC_KK_PIECES = 11;
constexpr uint32_t KK_FLAGS = 0, KK_EXPORT = 24, KK_PRE = 25, KK_A = 125, KK_C = 225, KK_CP = 545, KK_AP = 865, KK_APP = 2465,
                   KK_APP00 = 2565, KK_APPP00 = 2629, KK_INDEX = 2638, KK_IS_REAL = 2639;
// constraint numbering inside the hint (the reference's order of assertion)
constexpr uint32_t KK_J_FLAGS = 0, KK_J_SUM = 24, KK_J_INDEX = 25, KK_J_CP = 26, KK_J_A = 666, KK_J_DIFF = 2366, KK_J_APP = 2686,
                   KK_J_APP00 = 2786, KK_J_RC = 2854;

template <class F> KB_HD typename F::T zc_kk_xor(const typename F::T& a, const typename F::T& b) {
    const typename F::T ab = F::mul(a, b);
    return F::sub(F::add(a, b), F::add(ab, ab));
}
template <class F> KB_HD typename F::T zc_kk_bool(const typename F::T& a) { return F::mul(a, F::addc(a, kb::P - kb::R1)); }   // a (a - 1)

// ld(column relative to the chip's first Keccak column, owned) / sink(constraint index inside the hint, value)
template <class F, class Load, class Sink>
KB_HD void zc_keccak_piece(uint32_t q, Load&& ld, Sink&& sink) {
    using T = typename F::T;
    // bits of a limb in flight per lane: two for one-word values; one for extension values and 4-node vectors (4 words each: a
    // second bit costs ~60 VGPRs = a wave per SIMD, and the pieces are latency-bound: occupancy buys more than ILP)
    constexpr int KK_UNROLL = sizeof(T) > 4 ? 1 : 2;
    // rotation offsets r[x][y] (FIPS 202 section 3.2.2) and the round constants
    constexpr uint8_t R[25] = {0, 36, 3, 41, 18, 1, 44, 10, 45, 2, 62, 6, 43, 15, 61, 28, 55, 25, 21, 56, 27, 20, 39, 8, 14};
    constexpr uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
                                 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
                                 0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
                                 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    if (q == 0) {
        T sum = ld(KK_FLAGS, true);
        sink(KK_J_FLAGS, zc_kk_bool<F>(sum));
        T idx = F::mulc(sum, 0u);
#pragma unroll 1
        for (uint32_t i = 1; i < 24; i++) {
            const T f = ld(KK_FLAGS + i, true);
            sink(KK_J_FLAGS + i, zc_kk_bool<F>(f));
            sum = F::add(sum, f);
            idx = F::add(idx, F::mulc(f, kb::to_monty(i)));
        }
        sink(KK_J_SUM, F::addc(sum, kb::P - kb::R1));
        sink(KK_J_INDEX, F::mul(ld(KK_IS_REAL, false), F::sub(idx, ld(KK_INDEX, false))));
#pragma unroll 1
        for (uint32_t c = KK_EXPORT; c < KK_A; c++) (void)ld(c, true);          // export + preimage: no constraint reads them
#pragma unroll 1
        for (uint32_t limb = 0; limb < 4; limb++) {
            T acc = F::mulc(sum, 0u), accx = acc;
#pragma unroll 1
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t z = limb * 16 + 15 - k;
                const T bit = ld(KK_APP00 + z, true);
                sink(KK_J_APP00 + limb * 17 + k, zc_kk_bool<F>(bit));
                acc = F::add(F::add(acc, acc), bit);
                // A'''[0, 0, z] = A''[0, 0, z] xor (sum over the rounds r whose constant has bit z of flag r)
                T x = bit;
                if (z == 0 || z == 1 || z == 3 || z == 7 || z == 15 || z == 31 || z == 63) {        // the only bit positions any RC sets
                    T rc = F::mulc(bit, 0u);
#pragma unroll 1
                    for (uint32_t r = 0; r < 24; r++)
                        if ((RC[r] >> z) & 1) rc = F::add(rc, ld(KK_FLAGS + r, false));
                    x = zc_kk_xor<F>(bit, rc);
                }
                accx = F::add(F::add(accx, accx), x);
            }
            sink(KK_J_APP00 + limb * 17 + 16, F::sub(acc, ld(KK_APP + limb, false)));
            sink(KK_J_RC + limb, F::sub(accx, ld(KK_APPP00 + limb, true)));
        }
        return;
    }
    if (q <= 5) {
        // everything that lives in lane column x: C'[x, z] and the parity check, and — sharing the loads of C[x, .], C'[x, .] and
        // A'[., x, .] — the limbs of A[y, x] for all five y (xor3(A', C, C') = xor(A', xor(C, C')))
        const uint32_t x = q - 1, xm = (x + 4) % 5, xp = (x + 1) % 5;
#pragma unroll 1
        for (uint32_t limb = 0; limb < 4; limb++) {
            T acc[5];
#pragma unroll KK_UNROLL
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t z = limb * 16 + 15 - k;
                const T c = ld(KK_C + x * 64 + z, true);
                const T cpr = ld(KK_CP + x * 64 + z, true);
                sink(KK_J_CP + (x * 64 + z) * 2, zc_kk_bool<F>(c));
                const T inner = zc_kk_xor<F>(ld(KK_C + xm * 64 + z, false), ld(KK_C + xp * 64 + (z + 63) % 64, false));
                sink(KK_J_CP + (x * 64 + z) * 2 + 1, F::sub(cpr, zc_kk_xor<F>(c, inner)));
                const T cc = zc_kk_xor<F>(c, cpr);
                T d = F::sub(F::mulc(c, 0u), cpr);
#pragma unroll
                for (uint32_t y = 0; y < 5; y++) {
                    const T ap = ld(KK_AP + (y * 5 + x) * 64 + z, true);
                    sink(KK_J_A + ((y * 5 + x) * 4 + limb) * 17 + k, zc_kk_bool<F>(ap));
                    const T bit = zc_kk_xor<F>(ap, cc);
                    acc[y] = k == 0 ? bit : F::add(F::add(acc[y], acc[y]), bit);
                    d = F::add(d, ap);
                }
                sink(KK_J_DIFF + x * 64 + z, F::mul(F::mul(d, F::addc(d, kb::P - kb::to_monty(2))), F::addc(d, kb::P - kb::to_monty(4))));
            }
#pragma unroll
            for (uint32_t y = 0; y < 5; y++)
                sink(KK_J_A + ((y * 5 + x) * 4 + limb) * 17 + 16, F::sub(acc[y], ld(KK_A + (y * 5 + x) * 4 + limb, true)));
        }
        return;
    }
    {                                                  // the limbs of A''[y, .]: chi over the five lanes B[., y] = rho-pi(A'), each bit loaded once
        const uint32_t y = q - 6;
        uint32_t bcol[5], brot[5];
#pragma unroll
        for (uint32_t bx = 0; bx < 5; bx++) {          // B[bx, y, z] = A'[(bx + 3 y) % 5, bx][z - r]
            const uint32_t xa = (bx + 3 * y) % 5, ya = bx;
            bcol[bx] = KK_AP + (ya * 5 + xa) * 64;
            brot[bx] = 64 - R[xa * 5 + ya];
        }
#pragma unroll 1
        for (uint32_t limb = 0; limb < 4; limb++) {
            T acc[5];
#pragma unroll KK_UNROLL
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t z = limb * 16 + 15 - k;
                T bv[5];
#pragma unroll
                for (uint32_t bx = 0; bx < 5; bx++) bv[bx] = ld(bcol[bx] + (z + brot[bx]) % 64, false);
#pragma unroll
                for (uint32_t x = 0; x < 5; x++) {
                    const T& b1 = bv[(x + 1) % 5];
                    const T& b2 = bv[(x + 2) % 5];
                    const T bit = zc_kk_xor<F>(bv[x], F::sub(b2, F::mul(b1, b2)));
                    acc[x] = k == 0 ? bit : F::add(F::add(acc[x], acc[x]), bit);
                }
            }
#pragma unroll
            for (uint32_t x = 0; x < 5; x++)
                sink(KK_J_APP + (y * 5 + x) * 4 + limb, F::sub(acc[x], ld(KK_APP + (y * 5 + x) * 4 + limb, true)));
        }
    }
}

}  // namespace sp1hip
