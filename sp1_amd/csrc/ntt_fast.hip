// sp1_amd/csrc/ntt_fast.hip — register-radix passes of the batched RS-encode NTT for large
// transforms (2^14 <= N <= 2^24 per column). Same mathematics and output order as ntt.hip (one
// decimation-in-frequency transform of the zero-padded column, bit-reversed output, split into
// passes with an inter-pass twiddle); what changes is how a pass is executed on a CU:
//
//  * a workgroup (4 waves) owns a tile of 64 independent r-point transforms, r = 2^(A+B) in
//    {64, 128, 256}; LANE = transform, so every butterfly partner of an element lives in the same
//    lane and every stage twiddle is WAVE-UNIFORM: twiddles are scalar loads (s_load) into SGPRs,
//    they cost no VALU issue slots, no LDS traffic and no per-lane address arithmetic;
//  * the A+B stages run in two register steps (radix 2^A over the 2^B-strided elements, then radix
//    2^B over consecutive elements) with ONE in-place exchange through LDS between them instead of
//    one LDS round trip + barrier per stage;
//  * strided passes read/write 256 B runs (64 adjacent sub-transforms x 4 B); the last pass works
//    on contiguous runs and is transposed through a padded LDS tile so the lanes still index
//    independent transforms;
//  * the inter-pass twiddle w_seg^(i2 k1) comes from a precomputed table laid out like the segment itself (one
//    coalesced load next to the store; shared by all columns, so it is served by L2 / MALL): one multiplication per
//    element instead of building the twiddle from a per-workgroup and a per-lane factor (two).
// VALU work per butterfly is then just the field arithmetic: add (3) and the signed Montgomery product of
// the difference (7, sub_mul_tw); the radix steps whose base index is 0 skip the multiplications by 1.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "device_ctx.hpp"

namespace sp1hip {

constexpr int FT = 64;  // transforms per tile = lanes per wave

struct FastPassArgs {
    const uint32_t* in;      // FIRST: [cols][2^lg_n_in]; otherwise unused
    uint32_t* out;           // [cols][2^lg_total]
    const uint32_t* tw_r;    // w_r^j, j < r/2 (Montgomery), r = 2^(A+B)
    const uint32_t* tw_lane; // strided: [r][64] = w_seg^(k1 * c); contiguous: unused
    const uint32_t* tw_lo;
    const uint32_t* tw_hi;
    int lg_total, lg_seg, lg_n_in;
};

// (a - b) * w for canonical a, b and a wave-uniform twiddle given as the pair (w, w' = w p^-1 mod 2^32): the
// signed Montgomery product hi(d w) - hi((d w' mod 2^32) p) of d = a - b in (-p, p) is exact (both 64-bit
// products share their low word), lies in (-p, p), and takes v_sub + v_mul_lo + 2 v_mul_hi_i32 + v_sub;
// one add + unsigned-min brings it back to [0, p). One VALU slot less than widening a - b + p, multiplying
// 32 x 32 -> 64 and reducing with v_mul_lo + v_mad_u64 + correction.
__device__ __forceinline__ uint32_t sub_mul_tw(uint32_t a, uint32_t b, uint32_t w, uint32_t wm) {
    const int32_t d = (int32_t)(a - b);
    const int32_t q = (int32_t)((uint32_t)d * wm);
    const uint32_t r = (uint32_t)(__mulhi(d, (int32_t)w) - __mulhi(q, (int32_t)kb::P));
    return kb::umin(r, r + kb::P);
}

template <int K>
struct Radix {
    // In-register DIF over x[0 .. 2^K): element q sits at transform index i = i0 + q * 2^LG_STEP; the
    // stages handled are s = LG_STEP + K .. LG_STEP + 1 of an r = 2^LG_R point transform. Twiddle for
    // the butterfly (q, q + hq) at stage s: w_r^(((i0 + (q mod hq) 2^LG_STEP)) << (LG_R - s)); tw_r holds
    // w_r^j for j < r/2 followed by the matching w' words. ZERO_BASE: i0 is known to be 0, so the butterflies
    // with q mod hq == 0 have twiddle 1 (47 % of a radix-16 step) and skip the multiplication.
    // ZB: the upper (1 - 2^-ZB) of x[] is known to be zero (first pass of a zero-padded encode, blowup >= 2^ZB):
    // a butterfly whose partner is zero is a copy and one multiplication, one with both inputs zero vanishes. The
    // zero pattern is a compile-time bitmask that the fully unrolled loops fold away.
    template <int LG_R, int LG_STEP, bool ZERO_BASE, int ZB = 0>
    static __device__ __forceinline__ void run(uint32_t (&x)[1 << K], uint32_t i0, const uint32_t* __restrict__ tw_r) {
        uint32_t nz = (ZB > 0 && ZB <= K) ? ((1u << (1 << (K - ZB))) - 1u) : ((1 << K) == 32 ? ~0u : (1u << (1 << K)) - 1u);
#pragma unroll
        for (int t = K; t >= 1; t--) {
            const int hq = 1 << (t - 1);
            const int s = LG_STEP + t;
            uint32_t nz_next = nz;
#pragma unroll
            for (int q = 0; q < (1 << K); q++) {
                if (q & hq) continue;
                const bool a_nz = (nz >> q) & 1u, b_nz = (nz >> (q + hq)) & 1u;
                if (!a_nz && !b_nz) continue;
                const uint32_t a = x[q], b = x[q + hq];
                if (b_nz) x[q] = kb::add(a, b);
                if (ZERO_BASE && (q & (hq - 1)) == 0) {
                    x[q + hq] = b_nz ? kb::sub(a, b) : a;
                } else {
                    const uint32_t e = (i0 + (uint32_t)((q & (hq - 1)) << LG_STEP)) << (LG_R - s);
                    x[q + hq] = sub_mul_tw(a, b_nz ? b : 0u, tw_r[e], tw_r[e + (1u << (LG_R - 1))]);   // wave-uniform -> scalar loads
                }
                nz_next |= (1u << q) | (1u << (q + hq));
            }
            nz = nz_next;
        }
    }
};

// Pointers are separate __restrict__ kernel arguments (not a struct) so the compiler knows the
// twiddle tables cannot alias the output and keeps their wave-uniform loads on the scalar unit.
// WAVES: waves per workgroup. The 256-point tiles take ~65 KiB of LDS (two workgroups per CU), so they run with 8
// waves per workgroup to keep 16 waves per CU in flight; the 64/128-point tiles use 4.
// ZB (FIRST passes only): log2 of the zero-padding factor the radix step may rely on (0 or 2).
template <int A, int B, bool STRIDED, bool FIRST, int WAVES, int ZB = 0>
__global__ __launch_bounds__(64 * WAVES) void ntt_fast_pass(const uint32_t* __restrict__ a_in, uint32_t* __restrict__ a_out,
                                                     const uint32_t* __restrict__ a_tw_r,
                                                     const uint32_t* __restrict__ a_tw_lane,
                                                     const uint32_t* __restrict__ a_tw_lo,
                                                     const uint32_t* __restrict__ a_tw_hi, int a_lg_total, int a_lg_seg,
                                                     int a_lg_n_in) {
    struct {
        const uint32_t* __restrict__ in;
        uint32_t* __restrict__ out;
        const uint32_t* __restrict__ tw_r;
        const uint32_t* __restrict__ tw_lane;
        const uint32_t* __restrict__ tw_lo;
        const uint32_t* __restrict__ tw_hi;
        int lg_total, lg_seg, lg_n_in;
    } p{a_in, a_out, a_tw_r, a_tw_lane, a_tw_lo, a_tw_hi, a_lg_total, a_lg_seg, a_lg_n_in};
    constexpr int LG_R = A + B, R = 1 << LG_R;
    constexpr int PITCH = STRIDED ? FT : R + 1;   // contiguous tiles are [lane][i] with an odd pitch
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* tile = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is wave-uniform; tell the compiler so (keeps twiddle/table addresses in SGPRs)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // strided passes: the COLUMN is the fast grid index, so consecutive workgroups work on the same tile position of
    // different columns and share that tile's slice of the inter-pass twiddle table in their XCD's L2 (with 8 XCDs
    // taking workgroups round-robin, every XCD sees n_cols / 8 of them back to back)
    const uint32_t col = STRIDED ? blockIdx.x : blockIdx.y;
    const uint32_t tile_idx = STRIDED ? blockIdx.y : blockIdx.x;
    const uint64_t col_off = (uint64_t)col << p.lg_total;
    auto addr = [&](uint32_t i, uint32_t l) -> uint32_t { return STRIDED ? i * FT + l : l * PITCH + i; };

    uint32_t i2_0 = 0;
    uint64_t base = 0;
    const int lg_st = p.lg_seg - LG_R;
    if (STRIDED) {
        const uint32_t tiles_per_seg = 1u << (lg_st - 6);
        const uint32_t seg = tile_idx / tiles_per_seg;
        i2_0 = (tile_idx % tiles_per_seg) << 6;
        base = col_off + ((uint64_t)seg << p.lg_seg) + i2_0;
        // global -> LDS: row i is a 256 B run
        if (FIRST) {
            const uint32_t n_in = 1u << p.lg_n_in;
            const uint32_t* src = p.in + ((uint64_t)col << p.lg_n_in);
#pragma unroll 4
            for (uint32_t i = wave; i < R; i += WAVES) {
                const uint32_t idx = (i << lg_st) + i2_0 + lane;
                tile[i * FT + lane] = idx < n_in ? src[idx] : 0u;
            }
        } else {
#pragma unroll 4
            for (uint32_t i = wave; i < R; i += WAVES) tile[i * FT + lane] = p.out[base + ((uint64_t)i << lg_st) + lane];
        }
    } else {
        // 64 consecutive runs of R words; run `l` goes to LDS row l (pitch R + 1)
        base = col_off + ((uint64_t)tile_idx << (LG_R + 6));
        if (FIRST) {
            const uint32_t n_in = 1u << p.lg_n_in;
            const uint32_t* src = p.in + ((uint64_t)col << p.lg_n_in);
            const uint64_t off = (uint64_t)tile_idx << (LG_R + 6);
            for (uint32_t e = tid; e < (uint32_t)R * FT; e += 64 * WAVES)
                tile[(e >> LG_R) * PITCH + (e & (R - 1))] = (off + e) < n_in ? src[off + e] : 0u;
        } else {
            for (uint32_t e = tid; e < (uint32_t)R * FT; e += 64 * WAVES)
                tile[(e >> LG_R) * PITCH + (e & (R - 1))] = p.out[base + e];
        }
    }
    __syncthreads();

    // ---- step 1: radix 2^A over elements i = i_lo + q 2^B (stages LG_R .. B+1), in place in LDS
    for (uint32_t i_lo = wave; i_lo < (1u << B); i_lo += WAVES) {
        uint32_t x[1 << A];
#pragma unroll
        for (int q = 0; q < (1 << A); q++) x[q] = (ZB > 0 && q >= (1 << (A - ZB))) ? 0u : tile[addr(i_lo + ((uint32_t)q << B), lane)];
        Radix<A>::template run<LG_R, B, false, ZB>(x, i_lo, p.tw_r);
#pragma unroll
        for (int q = 0; q < (1 << A); q++) tile[addr(i_lo + ((uint32_t)q << B), lane)] = x[q];
    }
    __syncthreads();

    // ---- step 2: radix 2^B over consecutive elements i = i_hi 2^B + q (stages B .. 1)
    for (uint32_t i_hi = wave; i_hi < (1u << A); i_hi += WAVES) {
        uint32_t x[1 << B];
#pragma unroll
        for (int q = 0; q < (1 << B); q++) x[q] = tile[addr((i_hi << B) + q, lane)];
        Radix<B>::template run<LG_R, 0, true>(x, 0u, p.tw_r);
        if (STRIDED) {
#pragma unroll
            for (int q = 0; q < (1 << B); q++) {
                const uint32_t i = (i_hi << B) + q;
                // inter-pass twiddle w_seg^(i2 * bitrev(i)) from a table laid out like the segment itself (same
                // coalesced 256 B run as the store; shared by every column, so it lives in L2 / MALL)
                const uint64_t off = ((uint64_t)i << lg_st) + i2_0 + lane;
                p.out[base + ((uint64_t)i << lg_st) + lane] = kb::mul(x[q], p.tw_lane[off]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < (1 << B); q++) tile[addr((i_hi << B) + q, lane)] = x[q];
        }
    }
    if (!STRIDED) {
        __syncthreads();
        for (uint32_t e = tid; e < (uint32_t)R * FT; e += 64 * WAVES) p.out[base + e] = tile[(e >> LG_R) * PITCH + (e & (R - 1))];
    }
}

// tw_r tables (w_r^j, j < r/2, then the same entries times p^-1 mod 2^32) for r = 64, 128, 256 and lane tables per lg_seg, built on first use.
__global__ void fill_tw_r_kernel(uint32_t* out, int lg_r, const uint32_t* __restrict__ tw_lo,
                                 const uint32_t* __restrict__ tw_hi) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t half = (1u << lg_r) / 2;
    if (j >= half) return;
    const uint32_t e = j << (kb::TWO_ADICITY - lg_r);
    const uint32_t w = kb::mul(tw_hi[e >> TW_LO_BITS], tw_lo[e & (TW_LO - 1)]);
    out[j] = w;
    out[half + j] = w * kb::MU;      // w p^-1 mod 2^32 for the signed Montgomery product (sub_mul_tw)
}
// out[(i << lg_st) + i2] = w_seg^(i2 * bitrev_{lg_r}(i)): the inter-pass twiddle of every element of a segment
__global__ void fill_tw_lane_kernel(uint32_t* out, int lg_seg, int lg_r, const uint32_t* __restrict__ tw_lo,
                                    const uint32_t* __restrict__ tw_hi) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << lg_seg)) return;
    const int lg_st = lg_seg - lg_r;
    const uint32_t i = t >> lg_st, i2 = t & ((1u << lg_st) - 1);
    const uint32_t k1 = kb::reverse_bits_len(i, lg_r);
    const uint32_t e = (uint32_t)(((uint64_t)i2 * k1) & ((1u << lg_seg) - 1)) << (kb::TWO_ADICITY - lg_seg);
    out[t] = kb::mul(tw_hi[e >> TW_LO_BITS], tw_lo[e & (TW_LO - 1)]);
}

struct FastTables {
    uint32_t* tw_r[9] = {nullptr};      // index lg_r (6..8)
    uint32_t* tw_lane[25][9] = {{nullptr}};  // index [lg_seg][lg_r]: full inter-pass twiddle table of a segment (2^lg_seg words)
};
static std::mutex g_fast_mutex;
static FastTables* g_fast_tables[64] = {nullptr};

static int get_fast_tables(const DeviceCtx* ctx, hipStream_t s, int lg_r, int lg_seg, const uint32_t** tw_r,
                           const uint32_t** tw_lane) {
    std::lock_guard<std::mutex> lock(g_fast_mutex);
    SP1HIP_REQUIRE(ctx->device >= 0 && ctx->device < 64, "device index out of range");
    if (!g_fast_tables[ctx->device]) g_fast_tables[ctx->device] = new FastTables();
    FastTables* ft = g_fast_tables[ctx->device];
    if (!ft->tw_r[lg_r]) {
        SP1HIP_HIP(hipMalloc((void**)&ft->tw_r[lg_r], ((size_t)1 << lg_r) * 4));
        hipLaunchKernelGGL(fill_tw_r_kernel, dim3(1), dim3(256), 0, s, ft->tw_r[lg_r], lg_r, ctx->d_tw_lo, ctx->d_tw_hi);
        SP1HIP_LAUNCH_CHECK();
        SP1HIP_HIP(hipStreamSynchronize(s));   // other streams may use the table next
    }
    if (lg_seg >= 0 && !ft->tw_lane[lg_seg][lg_r]) {
        SP1HIP_HIP(hipMalloc((void**)&ft->tw_lane[lg_seg][lg_r], ((size_t)4) << lg_seg));
        hipLaunchKernelGGL(fill_tw_lane_kernel, dim3(((1u << lg_seg) + 255) / 256), dim3(256), 0, s, ft->tw_lane[lg_seg][lg_r], lg_seg,
                           lg_r, ctx->d_tw_lo, ctx->d_tw_hi);
        SP1HIP_LAUNCH_CHECK();
        SP1HIP_HIP(hipStreamSynchronize(s));
    }
    *tw_r = ft->tw_r[lg_r];
    *tw_lane = lg_seg >= 0 ? ft->tw_lane[lg_seg][lg_r] : nullptr;
    return SP1HIP_SUCCESS;
}

template <int A, int B>
static int launch_pass(FastPassArgs args, bool strided, bool first, bool zero_quarters, uint32_t tiles, uint32_t n_cols, hipStream_t s) {
    constexpr int R = 1 << (A + B);
    const size_t lds = strided ? ((size_t)R * FT) * 4 : ((size_t)FT * (R + 1)) * 4;
    const dim3 grid = strided ? dim3(n_cols, tiles) : dim3(tiles, n_cols);
    using Kern = void (*)(const uint32_t*, uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*,
                          int, int, int);
    constexpr int WAVES = (A + B == 8) ? 8 : 4;
    Kern kern = strided ? (first ? (zero_quarters ? ntt_fast_pass<A, B, true, true, WAVES, 2> : ntt_fast_pass<A, B, true, true, WAVES>)
                                 : ntt_fast_pass<A, B, true, false, WAVES>)
                        : (first ? ntt_fast_pass<A, B, false, true, WAVES> : ntt_fast_pass<A, B, false, false, WAVES>);
    if (lds > 48 * 1024)   // 256-point tiles need ~65 KiB of the CU's 160 KiB LDS
        SP1HIP_TRY(ensure_dynamic_lds((const void*)kern, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES), lds, s, args.in, args.out, args.tw_r, args.tw_lane, args.tw_lo, args.tw_hi,
                       args.lg_total, args.lg_seg, args.lg_n_in);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

// Returns false if the size is outside the fast path (caller falls back to the generic kernels).
bool ntt_fast_plan(int lg_total, int bits[3], int* n_passes) {
    if (lg_total >= 18 && lg_total <= 24) {
        const int base = lg_total / 3, rem = lg_total % 3;
        for (int i = 0; i < 3; i++) bits[i] = base + (i >= 3 - rem ? 1 : 0);
        if (const char* e = getenv("SP1HIP_NTT_PLAN")) {      // experiment: "877" etc.
            if (strlen(e) == 3 && (e[0] - '0') + (e[1] - '0') + (e[2] - '0') == lg_total) for (int i = 0; i < 3; i++) bits[i] = e[i] - '0';
        }
        *n_passes = 3;
        return true;
    }
    if (lg_total >= 14 && lg_total <= 16) {
        bits[0] = lg_total - 8;
        bits[1] = 8;
        *n_passes = 2;
        return true;
    }
    return false;
}

int ntt_fast_encode(uint32_t* d_out, const uint32_t* d_in, int lg_n, int lg_blowup, size_t n_cols, const DeviceCtx* ctx,
                    hipStream_t s) {
    const int lg_total = lg_n + lg_blowup;
    int bits[3], n_passes;
    if (!ntt_fast_plan(lg_total, bits, &n_passes)) return SP1HIP_ERROR_INVALID_ARGUMENT;
    int lg_seg = lg_total;
    for (int p = 0; p < n_passes; p++) {
        const bool first = p == 0, last = p == n_passes - 1;
        const int lg_r = bits[p];
        FastPassArgs a{};
        a.in = d_in;
        a.out = d_out;
        a.tw_lo = ctx->d_tw_lo;
        a.tw_hi = ctx->d_tw_hi;
        a.lg_total = lg_total;
        a.lg_seg = lg_seg;
        a.lg_n_in = lg_n;
        SP1HIP_TRY(get_fast_tables(ctx, s, lg_r, last ? -1 : lg_seg, &a.tw_r, &a.tw_lane));
        const uint32_t tiles = 1u << (lg_total - lg_r - 6);
        ScopedTimer t(p == 0 ? "ntt_pass0" : (p == 1 ? "ntt_pass1" : "ntt_pass2"), s);
        const bool zq = first && lg_blowup >= 2;       // three quarters of the first pass's input are padding zeros
        if (lg_r == 6) SP1HIP_TRY((launch_pass<3, 3>(a, !last, first, zq, tiles, (uint32_t)n_cols, s)));
        else if (lg_r == 7) SP1HIP_TRY((launch_pass<3, 4>(a, !last, first, zq, tiles, (uint32_t)n_cols, s)));
        else SP1HIP_TRY((launch_pass<4, 4>(a, !last, first, zq, tiles, (uint32_t)n_cols, s)));
        lg_seg -= lg_r;
    }
    return SP1HIP_SUCCESS;
}

}  // namespace sp1hip
