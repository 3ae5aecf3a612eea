// sp1_amd/csrc/ntt.hip — batched Reed–Solomon encode (zero-padded forward NTT, bit-reversed output)
// of column-major KoalaBear columns on gfx950.
//
// Replaces `CpuDftEncoder::encode_batch` -> `Dft::dft(.., log_blowup, BitReversed, 0)` ->
// `Radix2DitParallel::coset_dft_batch` with shift = 1
// (/root/reference/slop/crates/basefold-prover/src/encoder.rs:L22-L38,
//  /root/reference/slop/crates/dft/src/p3.rs:L11-L49): per column, the 2^lg_n coefficients are
// zero-padded to N = 2^(lg_n + lg_blowup), transformed with w = two_adic_generator(lg N), and row
// j of the result holds f(w^bitrev(j)).
//
// Design (not the reference's per-column host loop, and no separate LDE/bit-reverse kernels):
// one decimation-in-frequency transform of the zero-padded column, natural order in ->
// bit-reversed order out, so the output permutation costs nothing. The lg N index bits are split
// into 1..3 passes; every pass stages a tile through LDS and runs all of its butterfly stages there:
//   strided pass   tile [r][T]: r = 2^bits points with stride st, T >= 32 adjacent sub-transforms
//                  so every global access is a >= 128 B contiguous run; ends with the inter-pass
//                  twiddle w_seg^(i2 * k1) and writes back in place,
//   contiguous pass  the last bits: whole 2^bits-word runs, fully coalesced, no twiddle epilogue.
// The first pass reads the (4x smaller) input, supplies the zero padding in LDS and writes the output
// array; all later passes are in place. All columns of the batch go in one launch (grid.y).
// Algorithmic HBM traffic: 4 n (1 + 2^b) bytes per column; this design moves
// 4 n (1 + 2^b (2 passes - 1)) bytes (see DESIGN.md §Kernels for the roofline).
#include <cstdlib>

#include "device_ctx.hpp"

namespace sp1hip {

constexpr int NTT_THREADS = 256;
constexpr int NTT_TILE_WORDS = 8192;  // 32 KiB data tile (+ r/2 twiddles) per workgroup

__device__ __forceinline__ uint32_t tw_pow24(const uint32_t* __restrict__ tw_lo, const uint32_t* __restrict__ tw_hi,
                                             uint32_t e) {
    return kb::mul(tw_hi[e >> TW_LO_BITS], tw_lo[e & (TW_LO - 1)]);
}

// DIF butterfly stages over the r-point dimension of an LDS tile.
//   STRIDED:    tile is [r][T], element (i, c) at i*T + c, lg_inner = lg T
//   contiguous: tile is [S][r], element (seg, i) at seg*r + i, lg_inner = lg r (count = S * r / 2 butterflies)
template <bool STRIDED>
__device__ __forceinline__ void lds_dif_stages(uint32_t* tile, const uint32_t* tw, int lg_r, int lg_other, int tid) {
    if (lg_r == 0) return;  // 1-point transform (uniform across the workgroup)
    const uint32_t n_bfly = 1u << (lg_r - 1 + lg_other);
    for (int s = lg_r; s >= 1; s--) {
        const uint32_t half = 1u << (s - 1);
        for (uint32_t b = tid; b < n_bfly; b += NTT_THREADS) {
            uint32_t j, other;
            if (STRIDED) { other = b & ((1u << lg_other) - 1); j = b >> lg_other; }
            else { j = b & ((1u << (lg_r - 1)) - 1); other = b >> (lg_r - 1); }
            const uint32_t jj = j & (half - 1);
            const uint32_t i0 = ((j >> (s - 1)) << s) | jj;
            const uint32_t i1 = i0 + half;
            const uint32_t a0 = STRIDED ? (i0 << lg_other) + other : (other << lg_r) + i0;
            const uint32_t a1 = STRIDED ? (i1 << lg_other) + other : (other << lg_r) + i1;
            const uint32_t w = tw[jj << (lg_r - s)];
            const uint32_t x = tile[a0], y = tile[a1];
            tile[a0] = kb::add(x, y);
            tile[a1] = kb::mul(kb::sub(x, y), w);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void load_pass_twiddles(uint32_t* tw, int lg_r, const uint32_t* __restrict__ tw_lo,
                                                   const uint32_t* __restrict__ tw_hi, int tid) {
    // tw[j] = w_r^j, j < r/2
    for (uint32_t j = tid; j < (1u << lg_r) / 2; j += NTT_THREADS)
        tw[j] = tw_pow24(tw_lo, tw_hi, j << (kb::TWO_ADICITY - lg_r));
}

// One strided pass over sub-transforms ("segments") of length 2^lg_seg inside columns of length
// 2^lg_total. FIRST: the pass is the first of the whole transform (lg_seg == lg_total), reads
// `in` ([cols][2^lg_n_in]) with implicit zero padding, writes `out`.
template <bool FIRST>
__global__ __launch_bounds__(NTT_THREADS) void ntt_strided_pass(const uint32_t* __restrict__ in, uint32_t* out,
                                                                int lg_total, int lg_seg, int lg_r, int lg_t,
                                                                int lg_n_in, const uint32_t* __restrict__ tw_lo,
                                                                const uint32_t* __restrict__ tw_hi) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* tile = lds;
    uint32_t* tw = lds + (1u << (lg_r + lg_t));
    const int tid = threadIdx.x;
    const uint32_t col = blockIdx.y;
    const int lg_st = lg_seg - lg_r;                       // stride between the r points
    const uint32_t tiles_per_seg = 1u << (lg_st - lg_t);
    const uint32_t seg = blockIdx.x / tiles_per_seg;
    const uint32_t i2_0 = (blockIdx.x % tiles_per_seg) << lg_t;
    const uint64_t col_off = (uint64_t)col << lg_total;
    const uint64_t base = col_off + ((uint64_t)seg << lg_seg) + i2_0;
    const uint32_t count = 1u << (lg_r + lg_t);
    const uint32_t tmask = (1u << lg_t) - 1;

    load_pass_twiddles(tw, lg_r, tw_lo, tw_hi, tid);
    if (FIRST) {
        const uint32_t n_in = 1u << lg_n_in;
        const uint32_t* src = in + ((uint64_t)col << lg_n_in);
        for (uint32_t e = tid; e < count; e += NTT_THREADS) {
            const uint32_t idx = ((e >> lg_t) << lg_st) + i2_0 + (e & tmask);
            tile[e] = idx < n_in ? src[idx] : 0u;
        }
    } else {
        for (uint32_t e = tid; e < count; e += NTT_THREADS)
            tile[e] = out[base + ((uint64_t)(e >> lg_t) << lg_st) + (e & tmask)];
    }
    __syncthreads();
    lds_dif_stages<true>(tile, tw, lg_r, lg_t, tid);
    // slot i holds frequency k1 = bitrev_r(i) of the r-point transform; apply w_seg^(i2 * k1)
    const int sh = kb::TWO_ADICITY - lg_seg;
    for (uint32_t e = tid; e < count; e += NTT_THREADS) {
        const uint32_t i = e >> lg_t, c = e & tmask;
        const uint32_t k1 = kb::reverse_bits_len(i, lg_r);
        const uint32_t ex = ((i2_0 + c) * k1) << sh;
        out[base + ((uint64_t)i << lg_st) + c] = kb::mul(tile[e], tw_pow24(tw_lo, tw_hi, ex));
    }
}

// Last pass: contiguous runs of r = 2^lg_r words, S = 2^lg_s runs per workgroup.
template <bool FIRST>
__global__ __launch_bounds__(NTT_THREADS) void ntt_contig_pass(const uint32_t* __restrict__ in, uint32_t* out,
                                                               int lg_total, int lg_r, int lg_s, int lg_n_in,
                                                               const uint32_t* __restrict__ tw_lo,
                                                               const uint32_t* __restrict__ tw_hi) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* tile = lds;
    uint32_t* tw = lds + (1u << (lg_r + lg_s));
    const int tid = threadIdx.x;
    const uint32_t col = blockIdx.y;
    const uint32_t count = 1u << (lg_r + lg_s);
    const uint64_t off = (uint64_t)blockIdx.x << (lg_r + lg_s);
    uint32_t* dst = out + ((uint64_t)col << lg_total) + off;

    load_pass_twiddles(tw, lg_r, tw_lo, tw_hi, tid);
    if (FIRST) {
        const uint32_t n_in = 1u << lg_n_in;
        const uint32_t* src = in + ((uint64_t)col << lg_n_in);
        for (uint32_t e = tid; e < count; e += NTT_THREADS) tile[e] = (off + e) < n_in ? src[off + e] : 0u;
    } else {
        for (uint32_t e = tid; e < count; e += NTT_THREADS) tile[e] = dst[e];
    }
    __syncthreads();
    lds_dif_stages<false>(tile, tw, lg_r, lg_s, tid);
    for (uint32_t e = tid; e < count; e += NTT_THREADS) dst[e] = tile[e];
}

struct PassPlan {
    int n_passes;
    int bits[3];
};

static PassPlan plan_passes(int lg_total) {
    PassPlan p{};
    if (lg_total <= 11) { p.n_passes = 1; p.bits[0] = lg_total; return p; }
    int k = (lg_total + 7) / 8;
    if (k < 2) k = 2;
    p.n_passes = k;
    int base = lg_total / k, rem = lg_total % k;
    // remainder goes to the LAST (contiguous, always fully coalesced) passes
    for (int i = 0; i < k; i++) p.bits[i] = base + (i >= k - rem ? 1 : 0);
    return p;
}

static inline int ilog2_floor(uint32_t x) { int l = 0; while ((2u << l) <= x) l++; return l; }

// ntt_fast.hip
bool ntt_fast_plan(int lg_total, int bits[3], int* n_passes);
int ntt_fast_encode(uint32_t* d_out, const uint32_t* d_in, int lg_n, int lg_blowup, size_t n_cols, const DeviceCtx* ctx,
                    hipStream_t s);

}  // namespace sp1hip

using namespace sp1hip;

extern "C" int sp1hip_rs_encode_batch(uint32_t* d_out, const uint32_t* d_in, int lg_n, int lg_blowup, size_t n_cols,
                                      sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_n >= 0 && lg_blowup >= 0, "negative size");
    SP1HIP_REQUIRE(lg_n + lg_blowup <= kb::TWO_ADICITY, "lg_n + lg_blowup exceeds the field's two-adicity (24)");
    SP1HIP_REQUIRE(n_cols <= 65535, "at most 65535 columns per call");
    if (n_cols == 0) return SP1HIP_SUCCESS;
    SP1HIP_REQUIRE(d_out && d_in, "null buffer");
    SP1HIP_REQUIRE(d_out != d_in, "d_out must not alias d_in");
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    hipStream_t s = S(stream);
    const int lg_total = lg_n + lg_blowup;
    {
        // large transforms: register-radix passes (ntt_fast.hip); SP1HIP_NTT_GENERIC=1 forces the
        // generic LDS kernels below (kept for small sizes and as an A/B reference)
        static const bool force_generic = getenv("SP1HIP_NTT_GENERIC") != nullptr;
        int bits[3], np;
        if (!force_generic && ntt_fast_plan(lg_total, bits, &np))
            return ntt_fast_encode(d_out, d_in, lg_n, lg_blowup, n_cols, ctx, s);
    }
    const PassPlan plan = plan_passes(lg_total);
    int lg_seg = lg_total;
    for (int p = 0; p < plan.n_passes; p++) {
        ScopedTimer t(p == 0 ? "ntt_pass0" : (p == 1 ? "ntt_pass1" : "ntt_pass2"), s);
        const int lg_r = plan.bits[p];
        const bool first = p == 0, last = p == plan.n_passes - 1;
        if (!last) {
            const int lg_st = lg_seg - lg_r;
            int lg_t = ilog2_floor(NTT_TILE_WORDS) - lg_r;
            if (lg_t > lg_st) lg_t = lg_st;
            const uint32_t tiles = 1u << (lg_total - lg_r - lg_t);
            const size_t lds = ((size_t)(1u << (lg_r + lg_t)) + (1u << lg_r) / 2 + 1) * 4;
            dim3 grid(tiles, (uint32_t)n_cols);
            if (first)
                hipLaunchKernelGGL(ntt_strided_pass<true>, grid, dim3(NTT_THREADS), lds, s, d_in, d_out, lg_total, lg_seg,
                                   lg_r, lg_t, lg_n, ctx->d_tw_lo, ctx->d_tw_hi);
            else
                hipLaunchKernelGGL(ntt_strided_pass<false>, grid, dim3(NTT_THREADS), lds, s, d_in, d_out, lg_total,
                                   lg_seg, lg_r, lg_t, lg_n, ctx->d_tw_lo, ctx->d_tw_hi);
        } else {
            int lg_s = ilog2_floor(NTT_TILE_WORDS) - lg_r;
            if (lg_s < 0) lg_s = 0;
            if (lg_s > lg_total - lg_r) lg_s = lg_total - lg_r;
            const uint32_t tiles = 1u << (lg_total - lg_r - lg_s);
            const size_t lds = ((size_t)(1u << (lg_r + lg_s)) + (1u << lg_r) / 2 + 1) * 4;
            dim3 grid(tiles, (uint32_t)n_cols);
            if (first)
                hipLaunchKernelGGL(ntt_contig_pass<true>, grid, dim3(NTT_THREADS), lds, s, d_in, d_out, lg_total, lg_r,
                                   lg_s, lg_n, ctx->d_tw_lo, ctx->d_tw_hi);
            else
                hipLaunchKernelGGL(ntt_contig_pass<false>, grid, dim3(NTT_THREADS), lds, s, d_in, d_out, lg_total, lg_r,
                                   lg_s, lg_n, ctx->d_tw_lo, ctx->d_tw_hi);
        }
        SP1HIP_LAUNCH_CHECK();
        lg_seg -= lg_r;
    }
    return SP1HIP_SUCCESS;
}
