// sp1_amd/csrc/zc_poseidon2.hpp — the Poseidon2 permutation sub-AIR of the zerocheck as ONE fused evaluation.
//
// A third of a core shard's cells (the Global chip: one permutation per global interaction) and the heaviest chip of the
// recursion machine (Poseidon2WideDeg3) carry the same 163 constraints over the same 179 columns:
// `eval_external_round` for r = 0..7 and `eval_internal_rounds`
// (/root/reference/crates/hypercube/src/operations/poseidon2/air.rs:L66-L144) over a `Poseidon2Degree3Cols`
// (/root/reference/crates/hypercube/src/operations/poseidon2/permutation.rs:L44-L56). As interpreted bytecode they are
// ~3,800 instruction words per row pair and node whose operands make a round trip through the LDS register file each, with one
// scalar instruction of decode per vector instruction: 21 % of the VALU issue rate and 0.9 TB/s on the real-chip shard
// (profiles/r04_traffic.json). A caller's program may therefore carry a HINT pseudo-instruction (op 16: kind, first main
// column) in front of those constraints; the planner (zerocheck.hip) then drops the 163 asserts from the bytecode and emits
// nine self-contained pieces instead — external round q (q = 0..7) and the 20 internal rounds (q = 8) — evaluated here with the
// state in VGPRs: the sequential form of the reference, no register file, no decode. The hint changes nothing but speed: the
// SSA program stays the definition (the oracle and the verifier ignore op 16), the planner CHECKS the hint against the SSA
// on a pseudo-random row before trusting it, and the proof bytes are the same (tests compare them with the oracle's).
#pragma once
#include "kb31.hpp"
#include "poseidon2.hpp"

// A scheduling fence between the extension-field products of the fused pieces: left alone, the machine scheduler interleaves a
// dozen independent 16-multiply products for ILP and the pieces end up at 210-256 VGPRs (two waves per SIMD, or one); with one
// product in flight at a time the live set is what the algorithm needs (two septic operands, a group sum, an accumulator).
#if defined(__HIP_DEVICE_COMPILE__)
#define ZC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// a column index the optimiser cannot see through: a SECOND load of a column the piece has read before is then a real load
// instead of 28 registers kept alive across a septic product (the columns are in cache; the registers are what is scarce)
#define ZC_OPAQUE_ASM(v) asm volatile("" : "+s"(v))
#else
#define ZC_SCHED_FENCE() ((void)0)
#define ZC_OPAQUE_ASM(v) ((void)0)
#endif
__host__ __device__ __forceinline__ uint32_t zc_opaque(uint32_t v) { ZC_OPAQUE_ASM(v); return v; }

namespace sp1hip {

constexpr uint32_t ZC_HINT = 16;               // SSA pseudo-op: [16, kind, first main column]; defines no value
constexpr uint32_t ZC_HINT_POSEIDON2 = 1;
constexpr uint32_t ZC_P2_CONSTRAINTS = 163, ZC_P2_COLUMNS = 179, ZC_P2_PIECES = 9;
// column map of Poseidon2Degree3Cols
constexpr int ZC_P2_EXT = 0, ZC_P2_INT = 128, ZC_P2_S0 = 144, ZC_P2_OUT = 163;

struct P2Base {
    using T = uint32_t;
    static KB_HD T add(T a, T b) { return kb::add(a, b); }
    static KB_HD T sub(T a, T b) { return kb::sub(a, b); }
    static KB_HD T mul(T a, T b) { return kb::mul(a, b); }
    static KB_HD T addc(T a, uint32_t c) { return kb::add(a, c); }
    static KB_HD T mulc(T a, uint32_t c) { return kb::mul(a, c); }
};
struct P2Ext {
    using T = kb::Ext;
    static KB_HD T add(const T& a, const T& b) { return kb::ext_add(a, b); }
    static KB_HD T sub(const T& a, const T& b) { return kb::ext_sub(a, b); }
    static KB_HD T mul(const T& a, const T& b) { return kb::ext_mul(a, b); }
    static KB_HD T addc(T a, uint32_t c) { a.c[0] = kb::add(a.c[0], c); return a; }
    static KB_HD T mulc(const T& a, uint32_t c) { return kb::ext_mul_base(a, c); }
};

// external_linear_layer_mut (air.rs:L17-L45): circ(2 M4, M4, M4, M4) — additions only
template <class F> KB_HD void zc_p2_external_linear(typename F::T* s) {
    using T = typename F::T;
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        const T x0 = s[j], x1 = s[j + 1], x2 = s[j + 2], x3 = s[j + 3];
        const T t01 = F::add(x0, x1), t23 = F::add(x2, x3);
        const T t0123 = F::add(t01, t23);
        const T t01123 = F::add(t0123, x1), t01233 = F::add(t0123, x3);
        s[j] = F::add(t01123, t01);
        s[j + 1] = F::add(t01123, F::add(x2, x2));
        s[j + 2] = F::add(t01233, t23);
        s[j + 3] = F::add(t01233, F::add(x0, x0));
    }
    T sums[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sums[k] = F::add(F::add(s[k], s[k + 4]), F::add(s[k + 8], s[k + 12]));
#pragma unroll
    for (int j = 0; j < 16; j++) s[j] = F::add(s[j], sums[j & 3]);
}

// internal_linear_layer_mut (air.rs:L47-L66): s_i <- (sum + d_i s_i) 2^-32, d = [-2, 1, 2, 4, ..., 2^13, 2^15]. In Montgomery
// form the factor 2^-32 is the word 1 (mulc(x, 1) is one reduction) and d_i 2^-32 is the plain word d_i mod p.
template <class F> KB_HD void zc_p2_internal_linear(typename F::T* s) {
    using T = typename F::T;
    T sum = s[0];
#pragma unroll
    for (int i = 1; i < 16; i++) sum = F::add(sum, s[i]);
    const T sr = F::mulc(sum, 1u);
    s[0] = F::add(sr, F::mulc(s[0], kb::P - 2u));
#pragma unroll
    for (int i = 1; i < 16; i++) s[i] = F::add(sr, F::mulc(s[i], i == 15 ? (1u << 15) : (1u << (i - 1))));
}

// Piece q of the 163 constraints. ld(column, owned) returns the leaf value of a column of the permutation (0..178); `owned`
// marks the ONE load of that column, over all nine pieces, that also carries the GKR-opening batching term. sink(j, value):
// constraint j (0..162, the reference's order) evaluates to `value`.
template <class F, class RC, class Load, class Sink>
KB_HD void zc_p2_piece(uint32_t q, const RC* rc, Load&& ld, Sink&& sink) {
    using T = typename F::T;
    T s[16];
    if (q < 8) {
#pragma unroll
        for (int i = 0; i < 16; i++) { s[i] = ld(ZC_P2_EXT + 16 * q + i, true); if (i & 1) ZC_SCHED_FENCE(); }   // two columns' loads in flight
        if (q == 0) zc_p2_external_linear<F>(s);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const T x = F::addc(s[i], rc->ext[q][i]);
            s[i] = F::mul(F::mul(x, x), x);
            ZC_SCHED_FENCE();
        }
        zc_p2_external_linear<F>(s);
        ZC_SCHED_FENCE();
        const uint32_t nxt = q == 3 ? ZC_P2_INT : q == 7 ? ZC_P2_OUT : ZC_P2_EXT + 16 * (q + 1);
#pragma unroll
        for (int i = 0; i < 16; i++) { sink(16 * q + i, F::sub(ld(nxt + i, q == 7), s[i])); ZC_SCHED_FENCE(); }      // assert_eq(next_state[i], state[i])
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) { s[i] = ld(ZC_P2_INT + i, true); if (i & 1) ZC_SCHED_FENCE(); }
    T lane0 = s[0];
#pragma unroll 1
    for (int r = 0; r < 20; r++) {
        const T x = F::addc(lane0, rc->internal[r]);
        s[0] = F::mul(F::mul(x, x), x);
        zc_p2_internal_linear<F>(s);
        if (r < 19) {
            lane0 = ld(ZC_P2_S0 + r, true);                                                    // the next round starts from the COLUMN
            sink(128 + r, F::sub(lane0, s[0]));
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) { sink(147 + i, F::sub(ld(ZC_P2_EXT + 64 + i, false), s[i])); ZC_SCHED_FENCE(); }   // external_rounds_state[4]
}


// ---- the septic-curve constraints of the Global chip (operations/global_interaction.rs:L203-L208,
// operations/global_accumulation.rs:L83-L131): F_p^7 = F_p[z] / (z^7 - 3 z - 5), curve y^2 = x^3 + 45 x + 41 z^3
// (hypercube/src/septic_extension.rs:L307-L325, septic_curve.rs:L101-L113, L170-L188). Hints:
//   [16, 2, xy]                          the next 7 asserts are  y^2 - (x^3 + 45 x + 41 z^3)             x = cols xy..xy+6, y = xy+7..xy+13
//   [16, 3 | is_real << 8, xy | acc << 16]  the next 14 asserts are sum_checker_x (7) and is_real * sum_checker_y (7) for
//                                        p1 = (acc .. acc+13), p2 = (xy .. xy+13), p3 = (acc+14 .. acc+27)
constexpr uint32_t ZC_HINT_SEPTIC_CURVE = 2, ZC_HINT_SEPTIC_SUM = 3;

// res = a b in F_p^7 (SQUARE: a a with the symmetric products doubled): the 13 group sums T_s = sum_{i + j = s} a_i b_j are
// formed one after the other and folded into the 7 result coefficients at once (z^7 = 3 z + 5: T_s -> 5 res[s - 7] + 3 res[s - 6]),
// every index a compile-time constant: two operands, one result and one group sum are live, nothing is computed twice.
template <class F, bool SQUARE>
KB_HD void zc_septic_mul(const typename F::T* a, const typename F::T* b, typename F::T* res) {
    using T = typename F::T;
    const uint32_t c5 = kb::to_monty(5), c3 = kb::to_monty(3);
#pragma unroll
    for (int s_ = 0; s_ < 13; s_++) {
        T acc{};
        bool first = true;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const int j = s_ - i;
            if (j < 0 || j > 6) continue;
            if (SQUARE && i > j) continue;
            T t = F::mul(a[i], SQUARE ? a[j] : b[j]);
            if (SQUARE && i < j) t = F::add(t, t);
            acc = first ? t : F::add(acc, t);
            first = false;
            ZC_SCHED_FENCE();
        }
        if (s_ < 7) res[s_] = acc;
        else {
            res[s_ - 7] = F::add(res[s_ - 7], F::mulc(acc, c5));
            res[s_ - 6] = F::add(res[s_ - 6], F::mulc(acc, c3));
        }
    }
}

// piece 0 of kind 2: the curve equation. ld(column relative to xy, owned).
template <class F, class Load, class Sink>
KB_HD void zc_septic_curve_piece(Load&& ld, Sink&& sink) {
    using T = typename F::T;
    T x[7], t[7], u[7];
#pragma unroll
    for (int i = 0; i < 7; i++) x[i] = ld(i, true);
    zc_septic_mul<F, true>(x, x, t);                      // x^2
    zc_septic_mul<F, false>(t, x, u);                     // x^3
#pragma unroll
    for (int k = 0; k < 7; k++) {
        u[k] = F::add(u[k], F::mulc(x[k], kb::to_monty(45)));
        x[k] = ld(7 + k, true);                           // y takes x's registers
    }
    u[3] = F::addc(u[3], kb::to_monty(41));
    zc_septic_mul<F, true>(x, x, t);                      // y^2
#pragma unroll
    for (int k = 0; k < 7; k++) sink(k, F::sub(t[k], u[k]));
}

// pieces of kind 3: q = 0: sum_checker_x, q = 1: is_real * sum_checker_y. xy(col, owned) / acc(col, owned) / real().
template <class F, class LoadXY, class LoadAcc, class Real, class Sink>
KB_HD void zc_septic_sum_piece(uint32_t q, LoadXY&& xy, LoadAcc&& acc, Real&& real, Sink&& sink) {
    using T = typename F::T;
    T a[7], b[7], c[7];
    if (q == 0) {                                         // (p1.x + p2.x + p3.x) (p2.x - p1.x)^2 - (p2.y - p1.y)^2
#pragma unroll
        for (int i = 0; i < 7; i++) a[i] = F::sub(xy(i, false), acc(i, true));
        zc_septic_mul<F, true>(a, a, b);                  // dx^2
#pragma unroll
        for (int i = 0; i < 7; i++) a[i] = F::add(F::add(acc(i, false), xy(i, false)), acc(14 + i, true));
        zc_septic_mul<F, false>(a, b, c);
#pragma unroll
        for (int i = 0; i < 7; i++) a[i] = F::sub(xy(7 + i, false), acc(7 + i, true));
        zc_septic_mul<F, true>(a, a, b);                  // dy^2
#pragma unroll
        for (int k = 0; k < 7; k++) sink(k, F::sub(c[k], b[k]));
        return;
    }
    // is_real ((p1.y + p3.y) (p2.x - p1.x) - (p2.y - p1.y) (p1.x - p3.x))
#pragma unroll
    for (int i = 0; i < 7; i++) {
        a[i] = F::add(acc(7 + i, false), acc(21 + i, true));
        b[i] = F::sub(xy(i, false), acc(i, false));
    }
    zc_septic_mul<F, false>(a, b, c);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        a[i] = F::sub(xy(7 + i, false), acc(7 + i, false));
        b[i] = F::sub(acc(i, false), acc(14 + i, false));
    }
    T d[7];
    zc_septic_mul<F, false>(a, b, d);
    const T r = real();
#pragma unroll
    for (int k = 0; k < 7; k++) sink(7 + k, F::mul(r, F::sub(c[k], d[k])));
}


// ---- the septic pieces as the GPU runs them (round 5): weighted group sums instead of result coefficients.
// A septic product whose seven coefficients are constraints j0 .. j0 + 6 contributes sum_k alpha_{j0 + k} (a b)_k to the round's
// sum; with T_s = sum_{i + j = s} a_i b_j and z^7 = 3 z + 5 that is sum_s w_s T_s, w_s = alpha_{j0 + s} for s < 7 and
// 5 alpha_{j0 + s - 7} + 3 alpha_{j0 + s - 6} above — so the product is never materialised: two operands, one group sum and one
// accumulator are live (the result array and the seven alpha products of the per-coefficient form are gone), and a constraint's
// terms may come from DIFFERENT pieces (the sum over pieces is linear). Curve equation: 2 pieces (x^3 + 45 x + 41 z^3 / y^2), sum
// checkers: 4 (sx dx^2 / dy^2 / sy dx / dy px). The per-coefficient functions above stay the host model the planner checks hints
// with; these are what the kernels call (checked by every GPU proof against the oracle).
// K: KT<FIRST> of zc_device.hpp (K::scale(ext, T) = ext * T); alpha(j): wave-uniform alpha power of constraint j of the hint.
template <class F, class K, bool SQUARE, class Alpha>
__device__ __forceinline__ kb::Ext zc_septic_mul_weighted(const typename F::T* a, const typename F::T* b, Alpha&& alpha, uint32_t j0) {
    using T = typename F::T;
    const uint32_t c5 = kb::to_monty(5), c3 = kb::to_monty(3);
    kb::Ext out = kb::ext_zero();
#pragma unroll
    for (int s_ = 0; s_ < 13; s_++) {
        T acc{};
        bool first = true;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const int j = s_ - i;
            if (j < 0 || j > 6) continue;
            if (SQUARE && i > j) continue;
            T t = F::mul(a[i], SQUARE ? a[j] : b[j]);
            if (SQUARE && i < j) t = F::add(t, t);
            acc = first ? t : F::add(acc, t);
            first = false;
            ZC_SCHED_FENCE();
        }
        const kb::Ext w = s_ < 7 ? alpha(j0 + s_)
                                 : kb::ext_add(kb::ext_mul_base(alpha(j0 + s_ - 7), c5), kb::ext_mul_base(alpha(j0 + s_ - 6), c3));
        out = kb::ext_add(out, K::scale(w, acc));
        ZC_SCHED_FENCE();
    }
    return out;
}
template <class F, class K, class Load, class Alpha, class Emit>
__device__ __forceinline__ void zc_septic_curve_piece_w(uint32_t q, Load&& ld, Alpha&& alpha, Emit&& emit) {
    using T = typename F::T;
    T x[7];
    if (q == 1) {                                         // + y^2
#pragma unroll
        for (int i = 0; i < 7; i++) { x[i] = ld(7 + i, true); ZC_SCHED_FENCE(); }
        emit(zc_septic_mul_weighted<F, K, true>(x, x, alpha, 0));
        return;
    }
    T t[7];                                               // - (x^3 + 45 x + 41 z^3)
#pragma unroll
    for (int i = 0; i < 7; i++) { x[i] = ld(i, true); ZC_SCHED_FENCE(); }
    zc_septic_mul<F, true>(x, x, t);
    kb::Ext acc = zc_septic_mul_weighted<F, K, false>(t, x, alpha, 0);
#pragma unroll
    for (int k = 0; k < 7; k++) { acc = kb::ext_add(acc, K::scale(alpha(k), F::mulc(x[k], kb::to_monty(45)))); ZC_SCHED_FENCE(); }
    acc = kb::ext_add(acc, kb::ext_mul_base(alpha(3), kb::to_monty(41)));
    emit(kb::ext_sub(kb::ext_zero(), acc));
}
template <class F, class K, class LoadXY, class LoadAcc, class Real, class Alpha, class Emit>
__device__ __forceinline__ void zc_septic_sum_piece_w(uint32_t q, LoadXY&& xy, LoadAcc&& acc, Real&& real, Alpha&& alpha, Emit&& emit) {
    using T = typename F::T;
    T a[7], b[7];
    if (q == 0) {                                         // + (p1.x + p2.x + p3.x) (p2.x - p1.x)^2           constraints 0..6
#pragma unroll
        for (int i = 0; i < 7; i++) { a[i] = F::sub(xy(i, false), acc(i, true)); ZC_SCHED_FENCE(); }
        zc_septic_mul<F, true>(a, a, b);
#pragma unroll
        for (int i = 0; i < 7; i++) { a[i] = F::add(F::add(acc(zc_opaque(i), false), xy(zc_opaque(i), false)), acc(14 + i, true)); ZC_SCHED_FENCE(); }
        emit(zc_septic_mul_weighted<F, K, false>(a, b, alpha, 0));
        return;
    }
    if (q == 1) {                                         // - (p2.y - p1.y)^2
#pragma unroll
        for (int i = 0; i < 7; i++) { a[i] = F::sub(xy(7 + i, false), acc(7 + i, true)); ZC_SCHED_FENCE(); }
        emit(kb::ext_sub(kb::ext_zero(), zc_septic_mul_weighted<F, K, true>(a, a, alpha, 0)));
        return;
    }
    if (q == 2) {                                         // + is_real (p1.y + p3.y) (p2.x - p1.x)                    constraints 7..13
#pragma unroll
        for (int i = 0; i < 7; i++) {
            a[i] = F::add(acc(7 + i, false), acc(21 + i, true));
            b[i] = F::sub(xy(i, false), acc(i, false));
            ZC_SCHED_FENCE();
        }
        emit(K::scale(zc_septic_mul_weighted<F, K, false>(a, b, alpha, 7), real()));
        return;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {                         // - is_real (p2.y - p1.y) (p1.x - p3.x)
        a[i] = F::sub(xy(7 + i, false), acc(7 + i, false));
        b[i] = F::sub(acc(i, false), acc(14 + i, false));
        ZC_SCHED_FENCE();
    }
    emit(kb::ext_sub(kb::ext_zero(), K::scale(zc_septic_mul_weighted<F, K, false>(a, b, alpha, 7), real())));
}

}  // namespace sp1hip
