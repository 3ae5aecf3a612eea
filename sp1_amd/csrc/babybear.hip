// sp1_amd/csrc/babybear.hip — the commit path over BabyBear (BASELINE config 2: "NTT/LDE + Poseidon2 commit, both fields").
//
// The reference proves over KoalaBear (crates/primitives/src/lib.rs:L28) and keeps BabyBear as a second `IopCtx`
// (/root/reference/slop/crates/baby-bear/src/baby_bear_poseidon2.rs:L36-L55: DuplexChallenger / PaddingFreeSponge<16,8,8> /
// TruncatedPermutation over `Poseidon2<BabyBear, _, DiffusionMatrixBabyBear, 16, 7>`, 8 external + 13 internal rounds, L11-L30).
// This file is that instantiation for the two stages config 2 names: RS encode (`Dft::dft`, bit-reversed,
// slop/crates/dft/src/p3.rs:L11-L49) and the Poseidon2 Merkle commitment (`commit_tensors`, merkle-tree/src/p3sync.rs:L40-L143).
// p = 2^31 - 2^27 + 1, Montgomery words with R = 2^32 like KoalaBear's; same layouts (column-major tensors, leaf-first tree).
// PARITY: the internal diffusion matrix is the one parameter the reference tree does not restate — see oracle/bb_commit.hpp.
//
// Kernels (plain integer Montgomery arithmetic; the fp64 / lazy-reduction tuning of the KoalaBear leaf hash depends on that
// prime's bounds and is not carried over):
//   bb_ntt_pass     a pass = up to 8 DIF stages of 2^8-point sub-transforms, 16 adjacent ones per workgroup staged through
//                   LDS (every global access a 64 B run per row of the tile), twiddles from one table per transform size
//   bb_leaf_hash    one lane per row, the sponge state in VGPRs across all tensors' columns (coalesced column reads)
//   bb_compress     one lane per parent
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "device_ctx.hpp"
#include "tensor_table.hpp"

namespace sp1hip {
namespace bb {

constexpr uint32_t P = 0x78000001u;
constexpr uint32_t MU = 0x88000001u;             // p^-1 mod 2^32  (p MU = 1 mod 2^32)
constexpr int TWO_ADICITY = 27;
static_assert((uint32_t)(P * MU) == 1u, "Montgomery constant");

__host__ __device__ __forceinline__ uint32_t monty_reduce(uint64_t x) {       // x < 2^32 p  ->  x 2^-32 mod p
    const uint32_t t = (uint32_t)x * MU;
    const uint64_t u = (uint64_t)t * P;
    const uint32_t hi = (uint32_t)((x - u) >> 32);
    return x < u ? hi + P : hi;
}
__host__ __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { const uint32_t s = a + b; return s >= P ? s - P : s; }
__host__ __device__ __forceinline__ uint32_t sub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + P - b; }
__host__ __device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b) { return monty_reduce((uint64_t)a * b); }
inline uint32_t to_monty(uint32_t c) {
    const uint64_t r = ((uint64_t)1 << 32) % P;
    return monty_reduce((uint64_t)(c % P) * (uint32_t)((r * r) % P));
}
inline uint32_t pow(uint32_t b, uint64_t e) { uint32_t r = to_monty(1); while (e) { if (e & 1) r = mul(r, b); b = mul(b, b); e >>= 1; } return r; }
inline uint32_t two_adic_generator(int bits) {
    uint32_t g = pow(to_monty(31), (P - 1) >> TWO_ADICITY);
    for (int i = bits; i < TWO_ADICITY; i++) g = mul(g, g);
    return g;
}

struct RoundConstants { uint32_t ext[8][16], internal[13]; };
static const uint32_t RC_CANONICAL[30][16] = {
#include "bb_poseidon2_rc.inc"
};
inline RoundConstants make_round_constants() {
    RoundConstants rc;
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < 16; i++) { rc.ext[r][i] = to_monty(RC_CANONICAL[r][i]); rc.ext[4 + r][i] = to_monty(RC_CANONICAL[17 + r][i]); }
    for (int r = 0; r < 13; r++) rc.internal[r] = to_monty(RC_CANONICAL[4 + r][0]);
    return rc;
}

__device__ __forceinline__ void external_linear(uint32_t* s) {
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        const uint32_t x0 = s[j], x1 = s[j + 1], x2 = s[j + 2], x3 = s[j + 3];
        const uint32_t t01 = add(x0, x1), t23 = add(x2, x3), t0123 = add(t01, t23);
        const uint32_t t01123 = add(t0123, x1), t01233 = add(t0123, x3);
        s[j] = add(t01123, t01);
        s[j + 1] = add(t01123, add(x2, x2));
        s[j + 2] = add(t01233, t23);
        s[j + 3] = add(t01233, add(x0, x0));
    }
    uint32_t sums[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sums[k] = add(add(s[k], s[k + 4]), add(s[k + 8], s[k + 12]));
#pragma unroll
    for (int j = 0; j < 16; j++) s[j] = add(s[j], sums[j & 3]);
}
// s_i <- (sum + d_i s_i) 2^-32, d = [-2, 1, 2, 4, ..., 2^13, 2^15]: one 64-bit sum, one shift-add and one reduction per lane
__device__ __forceinline__ void internal_linear(uint32_t* s) {
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += s[i];
    const uint64_t v0 = s[0], neg0 = v0 ? P - v0 : 0;
    const uint32_t n0 = monty_reduce(sum - v0 + neg0);
#pragma unroll
    for (int i = 1; i < 16; i++) s[i] = monty_reduce(sum + ((uint64_t)s[i] << (i == 15 ? 15 : i - 1)));
    s[0] = n0;
}
__device__ __forceinline__ uint32_t sbox(uint32_t x) {
    const uint32_t x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2);
    return mul(x4, x3);
}
__device__ __forceinline__ void permute(uint32_t* s, const RoundConstants* __restrict__ rc) {
    external_linear(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox(add(s[i], rc->ext[r][i]));
        external_linear(s);
    }
#pragma unroll 1
    for (int r = 0; r < 13; r++) {
        s[0] = sbox(add(s[0], rc->internal[r]));
        internal_linear(s);
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox(add(s[i], rc->ext[r][i]));
        external_linear(s);
    }
}

__global__ __launch_bounds__(256) void bb_permute_kernel(uint32_t* __restrict__ states, size_t n, const RoundConstants* __restrict__ rc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = states[16 * i + k];
    permute(s, rc);
#pragma unroll
    for (int k = 0; k < 16; k++) states[16 * i + k] = s[k];
}

// leaf i = sponge over row i of all tensors' columns in message order (cols[c] = base pointer of column c, height rows)
__global__ __launch_bounds__(256) void bb_leaf_hash_kernel(const uint32_t* const* __restrict__ cols, uint32_t total_width, uint64_t height,
                                                           uint32_t* __restrict__ digests, const RoundConstants* __restrict__ rc) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= height) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = 0;
    uint32_t c = 0;
    for (; c + 8 <= total_width; c += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = cols[c + k][i];
        permute(s, rc);
    }
    if (c < total_width) {                       // a non-empty tail is absorbed and permuted (PaddingFreeSponge)
#pragma unroll
        for (int k = 0; k < 8; k++) if (c + k < total_width) s[k] = cols[c + k][i];
        permute(s, rc);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) digests[8 * i + k] = s[k];
}

__global__ __launch_bounds__(256) void bb_compress_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n_out,
                                                          const RoundConstants* __restrict__ rc) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = in[16 * i + k];
    permute(s, rc);
#pragma unroll
    for (int k = 0; k < 8; k++) out[8 * i + k] = s[k];
}

// root (8 words at `root`) + meta digest = hash([lg_height, total_width]) -> out[0..8) = root, out[8..16) = commitment
__global__ void bb_finalize_kernel(const uint32_t* __restrict__ root, uint32_t m_lg_h, uint32_t m_width, uint32_t* __restrict__ out,
                                   const RoundConstants* __restrict__ rc) {
    if (threadIdx.x != 0) return;
    uint32_t s[16];
    for (int k = 0; k < 16; k++) s[k] = 0;
    s[0] = m_lg_h; s[1] = m_width;
    permute(s, rc);
    uint32_t t[16];
    for (int k = 0; k < 8; k++) { t[k] = root[k]; t[8 + k] = s[k]; out[k] = root[k]; }
    permute(t, rc);
    for (int k = 0; k < 8; k++) out[8 + k] = t[k];
}

// ---- RS encode: zero-padded DIF NTT in place, natural order in, bit-reversed order out (column-major, one column per grid.y)
__global__ __launch_bounds__(256) void bb_pad_copy_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, uint64_t n, uint64_t N) {
    const uint64_t col = blockIdx.y;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256)
        out[col * N + i] = i < n ? in[col * n + i] : 0u;
}

// One pass: stages s_hi .. s_hi - b + 1 (stage s pairs elements 2^(s-1) apart). A workgroup owns a tile of 2^b (sub-transform)
// x 2^l (adjacent low indices) points of one column: point (m, q) of the tile is element hi_base + m 2^(s_hi - b) + lo_base + q.
constexpr int BB_TILE_LOG = 12;                  // 4096 words of LDS per workgroup
__global__ __launch_bounds__(256) void bb_ntt_pass_kernel(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw, int log_N, int s_hi, int b) {
    __shared__ uint32_t tile[1 << BB_TILE_LOG];
    const uint64_t N = (uint64_t)1 << log_N;
    const int low_bits = s_hi - b;                // bits below the sub-transform
    const int l = low_bits < BB_TILE_LOG - b ? low_bits : BB_TILE_LOG - b;
    const uint32_t pts = 1u << b, wq = 1u << l;
    uint32_t* col = data + (uint64_t)blockIdx.y * N;
    // tile index -> (high part above s_hi, low-part block)
    const uint64_t tiles_per_hi = (uint64_t)1 << (low_bits - l);
    const uint64_t hi = blockIdx.x / tiles_per_hi, lo_blk = blockIdx.x % tiles_per_hi;
    const uint64_t base = (hi << s_hi) + (lo_blk << l);
    const uint64_t mstride = (uint64_t)1 << low_bits;
    for (uint32_t e = threadIdx.x; e < pts * wq; e += 256) {
        const uint32_t m = e >> l, q = e & (wq - 1);
        tile[e] = col[base + m * mstride + q];
    }
    __syncthreads();
    for (int k = 0; k < b; k++) {                 // stage s = s_hi - k: half = 2^(b - 1 - k) in units of m
        const int s = s_hi - k;
        const uint32_t half_m = 1u << (b - 1 - k);
        for (uint32_t e = threadIdx.x; e < (pts >> 1) * wq; e += 256) {
            const uint32_t q = e & (wq - 1), bf = e >> l;                 // butterfly bf of column-slice q
            const uint32_t blk = bf / half_m, jm = bf % half_m;
            const uint32_t m0 = blk * 2 * half_m + jm, m1 = m0 + half_m;
            // position of the butterfly inside its stage-s block: j = (m0 mod 2^(...)) 2^low_bits + low index
            const uint64_t j = ((uint64_t)jm << low_bits) + (lo_blk << l) + q;
            const uint32_t t = tw[j << (log_N - s)];                      // w_N^(j N / 2^s)
            const uint32_t x = tile[(m0 << l) + q], y = tile[(m1 << l) + q];
            tile[(m0 << l) + q] = add(x, y);
            tile[(m1 << l) + q] = mul(sub(x, y), t);
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < pts * wq; e += 256) {
        const uint32_t m = e >> l, q = e & (wq - 1);
        col[base + m * mstride + q] = tile[e];
    }
}

__global__ __launch_bounds__(256) void bb_twiddle_kernel(uint32_t* __restrict__ tw, uint64_t half, uint32_t g, uint32_t one) {
    // tw[i] = g^i: each thread starts from g^(first index) by square-and-multiply, then walks 16 entries
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 >= half) return;
    uint32_t cur = one, b = g;
    for (uint64_t e = i0; e; e >>= 1) { if (e & 1) cur = mul(cur, b); b = mul(b, b); }
    for (uint64_t k = 0; k < 16 && i0 + k < half; k++) { tw[i0 + k] = cur; cur = mul(cur, g); }
}

struct BbCtx {
    RoundConstants* d_rc = nullptr;
    std::unordered_map<int, uint32_t*> twiddles;      // log_N -> table of N / 2 powers of w_N
    std::mutex m;
};
static int get_bb_ctx(BbCtx** out) {
    static std::mutex g;
    static std::unordered_map<int, BbCtx*> per_device;
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g);
    BbCtx*& c = per_device[dev];
    if (!c) {
        // published only when fully initialised: a failed allocation / copy must not leave a half-made context cached
        std::unique_ptr<BbCtx> fresh(new BbCtx());
        const RoundConstants rc = make_round_constants();
        SP1HIP_HIP(hipMalloc((void**)&fresh->d_rc, sizeof rc));
        hipError_t e = hipMemcpy(fresh->d_rc, &rc, sizeof rc, hipMemcpyHostToDevice);      // synchronous: complete on return
        if (e != hipSuccess) { (void)hipFree(fresh->d_rc); return map_hip_error(e, "uploading the BabyBear round constants"); }
        c = fresh.release();
    }
    *out = c;
    return SP1HIP_SUCCESS;
}
static int twiddles_for(BbCtx* c, int log_N, hipStream_t s, const uint32_t** out) {
    std::lock_guard<std::mutex> lk(c->m);
    auto it = c->twiddles.find(log_N);
    uint32_t* t = it == c->twiddles.end() ? nullptr : it->second;
    if (!t) {
        const uint64_t half = log_N ? ((uint64_t)1 << (log_N - 1)) : 1;
        SP1HIP_HIP(hipMalloc((void**)&t, half * 4));
        hipLaunchKernelGGL(bb_twiddle_kernel, dim3((unsigned)((half + 4095) / 4096)), dim3(256), 0, s, t, half, two_adic_generator(log_N), to_monty(1));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);      // the table is shared by later calls on any stream
        if (e != hipSuccess) { (void)hipFree(t); return map_hip_error(e, "building a BabyBear twiddle table"); }   // nothing cached on failure
        c->twiddles[log_N] = t;
    }
    *out = t;
    return SP1HIP_SUCCESS;
}

static int rs_encode(uint32_t* d_out, const uint32_t* d_in, int lg_n, int lg_blowup, size_t n_cols, hipStream_t s) {
    const int log_N = lg_n + lg_blowup;
    SP1HIP_REQUIRE(lg_n >= 0 && lg_blowup >= 0 && log_N <= TWO_ADICITY, "BabyBear transform size out of range");
    if (n_cols == 0) return SP1HIP_SUCCESS;
    SP1HIP_REQUIRE(d_out && d_in && d_out != d_in && n_cols < 65536, "bad buffers");
    const uint64_t n = (uint64_t)1 << lg_n, N = (uint64_t)1 << log_N;
    hipLaunchKernelGGL(bb_pad_copy_kernel, dim3((unsigned)std::min<uint64_t>((N + 255) / 256, 4096), (unsigned)n_cols), dim3(256), 0, s, d_out, d_in, n, N);
    SP1HIP_LAUNCH_CHECK();
    if (log_N == 0) return SP1HIP_SUCCESS;
    BbCtx* c;
    SP1HIP_TRY(get_bb_ctx(&c));
    const uint32_t* tw;
    SP1HIP_TRY(twiddles_for(c, log_N, s, &tw));
    for (int s_hi = log_N; s_hi >= 1;) {
        const int b = s_hi % 8 ? s_hi % 8 : 8;    // the short pass first: the last passes have no low bits to widen their tiles with
        const int low_bits = s_hi - b, l = std::min(low_bits, BB_TILE_LOG - b);
        const uint64_t tiles = N >> (b + l);
        hipLaunchKernelGGL(bb_ntt_pass_kernel, dim3((unsigned)tiles, (unsigned)n_cols), dim3(256), 0, s, d_out, tw, log_N, s_hi, b);
        SP1HIP_LAUNCH_CHECK();
        s_hi -= b;
    }
    return SP1HIP_SUCCESS;
}

static int merkle_commit(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, uint32_t* d_tree, uint32_t* d_root_and_commit,
                         hipStream_t s) {
    SP1HIP_REQUIRE(lg_height >= 0 && lg_height <= 30 && d_tree && d_root_and_commit, "bad arguments");
    TensorTable tab;
    uint32_t total_width = 0;
    SP1HIP_TRY(make_tensor_table(tensors, n_tensors, &tab, &total_width));
    const uint64_t h = (uint64_t)1 << lg_height;
    BbCtx* c;
    SP1HIP_TRY(get_bb_ctx(&c));
    const uint32_t** d_cols = nullptr;
    SP1HIP_TRY(arena_alloc((void**)&d_cols, std::max<size_t>(total_width, 1) * sizeof(uint32_t*), s));
    SP1HIP_TRY(expand_columns_async(tab, total_width, h, d_cols, s));
    hipLaunchKernelGGL(bb_leaf_hash_kernel, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, s, d_cols, total_width, h, d_tree, c->d_rc);
    SP1HIP_LAUNCH_CHECK();
    arena_free(d_cols, std::max<size_t>(total_width, 1) * sizeof(uint32_t*), s);
    uint64_t off = 0;
    for (uint64_t n = h; n > 1; n /= 2) {        // layers leaf-first, back to back
        hipLaunchKernelGGL(bb_compress_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, d_tree + off * 8, d_tree + (off + n) * 8, n / 2, c->d_rc);
        SP1HIP_LAUNCH_CHECK();
        off += n;
    }
    hipLaunchKernelGGL(bb_finalize_kernel, dim3(1), dim3(64), 0, s, d_tree + off * 8, to_monty((uint32_t)lg_height), to_monty(total_width),
                       d_root_and_commit, c->d_rc);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

}  // namespace bb
}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_bb_rs_encode_batch(uint32_t* d_out, const uint32_t* d_in, int lg_n, int lg_blowup, size_t n_cols, sp1hip_stream_t stream) {
    return bb::rs_encode(d_out, d_in, lg_n, lg_blowup, n_cols, S(stream));
}

int sp1hip_bb_merkle_commit(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, uint32_t* d_tree, uint32_t* d_root_and_commit,
                            sp1hip_stream_t stream) {
    return bb::merkle_commit(tensors, n_tensors, lg_height, d_tree, d_root_and_commit, S(stream));
}

int sp1hip_bb_commit_mles(const sp1hip_tensor_t* mles, int n_mles, int lg_n, int lg_blowup, uint32_t* const* d_codewords, uint32_t* d_tree,
                          uint32_t h_commit[8], sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(mles && n_mles > 0 && d_codewords && d_tree && h_commit, "null argument");
    std::vector<sp1hip_tensor_t> cws(n_mles);
    for (int k = 0; k < n_mles; k++) {
        SP1HIP_REQUIRE(d_codewords[k], "null codeword buffer");
        SP1HIP_TRY(bb::rs_encode(d_codewords[k], mles[k].d_data, lg_n, lg_blowup, mles[k].width, S(stream)));
        cws[k] = sp1hip_tensor_t{d_codewords[k], mles[k].width};
    }
    uint32_t* d_rc16 = nullptr;
    SP1HIP_TRY(arena_alloc((void**)&d_rc16, 64, S(stream)));
    int st = bb::merkle_commit(cws.data(), n_mles, lg_n + lg_blowup, d_tree, d_rc16, S(stream));
    uint32_t h16[16];
    if (st == SP1HIP_SUCCESS) {
        hipError_t e = hipMemcpyAsync(h16, d_rc16, 64, hipMemcpyDeviceToHost, S(stream));
        if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
        if (e != hipSuccess) st = map_hip_error(e, "sp1hip_bb_commit_mles");
    }
    arena_free(d_rc16, 64, S(stream));
    SP1HIP_TRY(st);
    for (int k = 0; k < 8; k++) h_commit[k] = h16[8 + k];
    return SP1HIP_SUCCESS;
}

int sp1hip_bb_poseidon2_permute(uint32_t* d_states, size_t n, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(d_states || n == 0, "null states");
    if (!n) return SP1HIP_SUCCESS;
    bb::BbCtx* c;
    SP1HIP_TRY(bb::get_bb_ctx(&c));
    hipLaunchKernelGGL(bb::bb_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), d_states, n, c->d_rc);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

}  // extern "C"
