// sp1_amd/csrc/basefold.hip — streaming kernels of the BaseFold opening on gfx950: column RLC,
// codeword / mle folds, eq tables, column evaluations, fold-round leaf hashing and openings, layout
// helpers.
//
//   batch            `FriCpuProver::batch`            /root/reference/slop/crates/basefold-prover/src/fri.rs:L31-L80
//   fold_even_odd    `p3_fri::fold_even_odd` (fri.rs:L118), formula pinned by
//                    /root/reference/slop/crates/basefold/src/verifier.rs:L364-L374
//   fold_mle         /root/reference/slop/crates/multilinear/src/fold.rs:L12-L26
//   partial_lagrange /root/reference/slop/crates/multilinear/src/lagrange.rs:L19-L45
//   mle_eval_columns /root/reference/slop/crates/multilinear/src/eval.rs:L9-L21
//   fixed_at_zero    /root/reference/slop/crates/multilinear/src/restrict.rs:L75-L87
//   pair leaves      fri.rs:L103-L108 (codeword reshaped [N/2][8]) + p3sync.rs leaf hashing
//
// All of these are single-pass streaming kernels: one lane per output row, every column access
// coalesced, ext vectors in SoA (4 base columns) so a wave reads 4 x 256 B contiguous runs.
#include "device_ctx.hpp"
#include "tensor_table.hpp"

namespace sp1hip {

struct ExtArg { uint32_t c[4]; };
__device__ __forceinline__ kb::Ext E(const ExtArg& a) { return kb::Ext{{a.c[0], a.c[1], a.c[2], a.c[3]}}; }

// ---------------------------------------------------------------- layout helpers
// 32x32 LDS-tiled transpose, padded against bank conflicts.
__global__ __launch_bounds__(256) void transpose_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                        size_t in_rows, size_t in_cols) {
    // in: [in_rows][in_cols] (in_cols fastest) -> out: [in_cols][in_rows]
    __shared__ uint32_t tile[32][33];
    const size_t r0 = (size_t)blockIdx.y * 32, c0 = (size_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        size_t r = r0 + k, c = c0 + tx;
        if (r < in_rows && c < in_cols) tile[k][tx] = in[r * in_cols + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        size_t c = c0 + k, r = r0 + tx;
        if (r < in_rows && c < in_cols) out[c * in_rows + r] = tile[tx][k];
    }
}

template <bool TO>
__global__ void monty_convert_kernel(uint32_t* data, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) data[i] = TO ? kb::to_monty(data[i]) : kb::from_monty(data[i]);
}

// ---------------------------------------------------------------- batch (RLC of all columns)
// out[r] = sum_g coeff[g] * col_g[r]. Coefficients are wave-uniform (scalar loads); the sum over the columns is
// accumulated unreduced (kb::DotAcc, total_width <= 2^16).
__global__ __launch_bounds__(256) void batch_kernel(const uint32_t* const* __restrict__ cols, uint32_t total_width,
                                                    uint32_t height, const uint32_t* __restrict__ coeffs,
                                                    uint32_t* __restrict__ out) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= height) return;
    kb::DotAcc acc;                   // delayed reduction: one Montgomery reduction per coordinate for the whole row
    kb::dot_init(acc);
    for (uint32_t g = 0; g < total_width; g++) {
        const kb::Ext c{{coeffs[4 * g], coeffs[4 * g + 1], coeffs[4 * g + 2], coeffs[4 * g + 3]}};   // wave-uniform
        kb::dot_add(acc, c, gptr(cols[g])[row]);
    }
    const kb::Ext r = kb::dot_finish(acc);
#pragma unroll
    for (int k = 0; k < 4; k++) out[(size_t)k * height + row] = r.c[k];
}

struct FoldOpenDesc {            // one fold round of the query phase (open_fold_rounds_kernel); vals_off / paths_off in words
    const uint32_t* cw;
    const uint32_t* tree;
    uint32_t lg_c, vals_off, paths_off, pad;
};

// ---------------------------------------------------------------- folds
// out[i] = (e0 + e1)/2 + beta * (e0 - e1) / (2 x_i),  x_i = w_N^{bitrev_{lg N}(2 i)}
__device__ __forceinline__ void fold_even_odd_at(const uint32_t* __restrict__ cw, int lg_n, const ExtArg& half_beta,
                                                 const uint32_t* __restrict__ tw_lo, const uint32_t* __restrict__ tw_hi,
                                                 uint32_t* __restrict__ out, uint32_t i) {
    const uint32_t n = 1u << lg_n, m = n >> 1;
    kb::Ext e0, e1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint2 v = *reinterpret_cast<const uint2*>(cw + (size_t)k * n + 2 * (size_t)i);
        e0.c[k] = v.x;
        e1.c[k] = v.y;
    }
    // 1 / x_i = w_N^(N - br), br = bitrev_{lg n}(2 i) < N/2
    const uint32_t br = kb::reverse_bits_len(2 * i, lg_n);
    const uint32_t ex = ((n - br) & (n - 1)) << (kb::TWO_ADICITY - lg_n);
    const uint32_t xinv = kb::mul(tw_hi[ex >> TW_LO_BITS], tw_lo[ex & (TW_LO - 1)]);
    const uint32_t inv2 = 0x00ffffffu;   // to_monty(1/2) = R1 / 2 (R1 = 2^25 - 2 is even)
    kb::Ext s = kb::ext_mul_base(kb::ext_add(e0, e1), inv2);
    kb::Ext d = kb::ext_mul_base(kb::ext_sub(e0, e1), xinv);
    kb::Ext r = kb::ext_add(s, kb::ext_mul(d, E(half_beta)));      // the challenge is wave-uniform: second operand
#pragma unroll
    for (int k = 0; k < 4; k++) out[(size_t)k * m + i] = r.c[k];
}
__global__ __launch_bounds__(256) void fold_even_odd_kernel(const uint32_t* __restrict__ cw, int lg_n, ExtArg half_beta,
                                                            const uint32_t* __restrict__ tw_lo,
                                                            const uint32_t* __restrict__ tw_hi,
                                                            uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < (1u << (lg_n - 1))) fold_even_odd_at(cw, lg_n, half_beta, tw_lo, tw_hi, out, i);
}

// out[i] = m[2 i] + beta m[2 i + 1]
__device__ __forceinline__ kb::Ext fold_mle_at(const uint32_t* __restrict__ mle, int lg_n, const ExtArg& beta,
                                               uint32_t* __restrict__ out, uint32_t i) {
    const uint32_t n = 1u << lg_n, m = n >> 1;
    kb::Ext a, b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint2 v = *reinterpret_cast<const uint2*>(mle + (size_t)k * n + 2 * (size_t)i);
        a.c[k] = v.x;
        b.c[k] = v.y;
    }
    kb::Ext r = kb::ext_add(a, kb::ext_mul(b, E(beta)));
#pragma unroll
    for (int k = 0; k < 4; k++) out[(size_t)k * m + i] = r.c[k];
    return r;
}
__global__ __launch_bounds__(256) void fold_mle_kernel(const uint32_t* __restrict__ mle, int lg_n, ExtArg beta,
                                                       uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < (1u << (lg_n - 1))) (void)fold_mle_at(mle, lg_n, beta, out, i);
}

__device__ __forceinline__ void block_sum4(uint32_t (&v)[4], uint32_t* scratch);

// One BaseFold commit-phase round's folds in ONE launch (prover.hip): workgroups [0, cw_blocks) fold the codeword, the
// rest fold the message AND leave the partial sums of the NEXT round's first univariate value,
// zero_val' = sum_j eq'[j] mle'[2 j] (eq' = the prefix table of one coordinate less), one ext per workgroup in
// `partial` — the folded entry is in a register at that point, so the next round needs neither a pass over the new
// message nor a launch for it. eq_next == nullptr: the message is down to one entry, no next round.
__global__ __launch_bounds__(256) void fold_round_kernel(const uint32_t* __restrict__ cw, int lg_c, ExtArg half_beta,
                                                         const uint32_t* __restrict__ tw_lo, const uint32_t* __restrict__ tw_hi,
                                                         uint32_t* __restrict__ cw_out, uint32_t cw_blocks,
                                                         const uint32_t* __restrict__ mle, int lg_m, ExtArg beta,
                                                         uint32_t* __restrict__ mle_out, const uint32_t* __restrict__ eq_next,
                                                         uint32_t* __restrict__ partial) {
    __shared__ uint32_t scratch[16];
    if (blockIdx.x < cw_blocks) {
        const uint32_t i = blockIdx.x * 256u + threadIdx.x;
        if (i < (1u << (lg_c - 1))) fold_even_odd_at(cw, lg_c, half_beta, tw_lo, tw_hi, cw_out, i);
        return;
    }
    const uint32_t b = blockIdx.x - cw_blocks, i = b * 256u + threadIdx.x, m = 1u << (lg_m - 1);
    kb::Ext term = kb::ext_zero();
    if (i < m) {
        const kb::Ext r = fold_mle_at(mle, lg_m, beta, mle_out, i);
        if (eq_next && !(i & 1u)) {
            const uint32_t j = i >> 1, half = m >> 1;
            kb::Ext e;
#pragma unroll
            for (int k = 0; k < 4; k++) e.c[k] = eq_next[(size_t)k * half + j];
            term = kb::ext_mul(e, r);
        }
    }
    if (!eq_next) return;
    block_sum4(term.c, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) partial[b * 4 + k] = term.c[k];
    }
}

// ---------------------------------------------------------------- eq tables
struct PointArg {
    uint32_t c[kb::TWO_ADICITY + 8][4];
};

// small[i] = prod_j (bit_j(i) ? x_j : 1 - x_j) over coordinates [first, first + d), bit_j big-endian.
// AoS ext output, 2^d entries.
__global__ void eq_small_kernel(PointArg pt, int first, int d, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << d)) return;
    kb::Ext acc = kb::ext_one();
    for (int j = 0; j < d; j++) {
        kb::Ext x{{pt.c[first + j][0], pt.c[first + j][1], pt.c[first + j][2], pt.c[first + j][3]}};
        const bool bit = (i >> (d - 1 - j)) & 1u;
        acc = kb::ext_mul(acc, bit ? x : kb::ext_sub(kb::ext_one(), x));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * (size_t)i + k] = acc.c[k];
}

// full[i] = hi[i >> d_lo] * lo[i & (2^d_lo - 1)]; SoA output of length 2^dim
__global__ __launch_bounds__(256) void eq_outer_kernel(const uint32_t* __restrict__ hi, const uint32_t* __restrict__ lo,
                                                       int d_lo, uint32_t len, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= len) return;
    const uint32_t ih = i >> d_lo, il = i & ((1u << d_lo) - 1);
    kb::Ext a{{hi[4 * ih], hi[4 * ih + 1], hi[4 * ih + 2], hi[4 * ih + 3]}};
    kb::Ext b{{lo[4 * il], lo[4 * il + 1], lo[4 * il + 2], lo[4 * il + 3]}};
    kb::Ext r = kb::ext_mul(a, b);
#pragma unroll
    for (int k = 0; k < 4; k++) out[(size_t)k * len + i] = r.c[k];
}

// Every prefix table of eq over the first t coordinates, t = 0..d, in one launch: table t is an ext SoA of length 2^t at
// word offset 4 (2^t - 1). Thread i walks the d coordinates of index i MSB-first and leaves the running product behind
// at every depth where the remaining bits of i are zero, so the 2^d threads do d products each and no table is a
// launch of its own (the BaseFold rounds read table lg_m - 1, prover.hip).
__global__ __launch_bounds__(256) void eq_prefix_soa_kernel(PointArg pt, int d, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (1u << d)) return;
    kb::Ext acc = kb::ext_one();
    for (int t = 0;; t++) {
        const int rest = d - t;
        if ((i & ((1u << rest) - 1u)) == 0) {
            const uint32_t len = 1u << t, idx = i >> rest;
            uint32_t* tab = out + 4 * ((size_t)len - 1);
#pragma unroll
            for (int k = 0; k < 4; k++) tab[(size_t)k * len + idx] = acc.c[k];
        }
        if (t == d) break;
        kb::Ext x{{pt.c[t][0], pt.c[t][1], pt.c[t][2], pt.c[t][3]}};
        const bool bit = (i >> (rest - 1)) & 1u;
        acc = kb::ext_mul(acc, bit ? x : kb::ext_sub(kb::ext_one(), x));
    }
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, __shfl_xor(v, off));
    return v;
}

// Block-reduces 4 words per lane; lane 0 of wave 0 returns the total (others undefined).
__device__ __forceinline__ void block_sum4(uint32_t (&v)[4], uint32_t* scratch /* [4][4] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) scratch[wave * 4 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            v[k] = kb::add(kb::add(scratch[k], scratch[4 + k]), kb::add(scratch[8 + k], scratch[12 + k]));
    }
    __syncthreads();
}

constexpr int EVAL_COLS = 4;      // columns per workgroup (4 x 16 VGPRs of unreduced accumulators)
constexpr int EVAL_ROWS = 16384;  // rows per workgroup

// partial[chunk][g] = sum over the chunk's rows of eq[r] * col_g[r]
__global__ __launch_bounds__(256) void eval_columns_partial_kernel(const uint32_t* const* __restrict__ cols,
                                                                   uint32_t total_width, uint32_t height,
                                                                   const uint32_t* __restrict__ eq,
                                                                   uint32_t* __restrict__ partial) {
    __shared__ uint32_t scratch[16];
    const uint32_t g0 = blockIdx.x * EVAL_COLS;      // consecutive workgroups share the rows (and their eq slice)
    const uint32_t r0 = blockIdx.y * EVAL_ROWS;
    kb::DotAcc dacc[EVAL_COLS];       // delayed reduction: EVAL_ROWS / 256 terms per lane, one reduction per column
#pragma unroll
    for (int c = 0; c < EVAL_COLS; c++) kb::dot_init(dacc[c]);
    for (uint32_t r = r0 + threadIdx.x; r < r0 + EVAL_ROWS && r < height; r += 256) {
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = eq[(size_t)k * height + r];
#pragma unroll
        for (int c = 0; c < EVAL_COLS; c++)
            if (g0 + c < total_width) kb::dot_add(dacc[c], e, gptr(cols[g0 + c])[r]);
    }
    uint32_t acc[EVAL_COLS][4];
#pragma unroll
    for (int c = 0; c < EVAL_COLS; c++) {
        const kb::Ext v = kb::dot_finish(dacc[c]);
#pragma unroll
        for (int k = 0; k < 4; k++) acc[c][k] = v.c[k];
    }
#pragma unroll
    for (int c = 0; c < EVAL_COLS; c++) {
        block_sum4(acc[c], scratch);
        if (threadIdx.x == 0 && g0 + c < total_width) {
#pragma unroll
            for (int k = 0; k < 4; k++) partial[((size_t)blockIdx.y * total_width + g0 + c) * 4 + k] = acc[c][k];
        }
    }
}

// out[j] = sum_chunk partial[chunk][j], j < n_words. A workgroup owns JW consecutive words and walks the chunks with
// 256 / JW lanes per word (a wave still reads 64 consecutive words of `partial`), then folds the lanes through LDS: the
// one-lane-per-word loop it replaces was a chain of n_chunks dependent adds behind as many loads (46 us for 512 chunks,
// once per BaseFold round).
template <int JW>
__global__ __launch_bounds__(256) void sum_partials_kernel(const uint32_t* __restrict__ partial, uint32_t n_chunks, uint32_t n_words,
                                                           uint32_t* __restrict__ out) {
    constexpr uint32_t CL = 256 / JW;
    __shared__ uint32_t sm[256];
    const uint32_t jl = threadIdx.x % JW, cl = threadIdx.x / JW, j = blockIdx.x * JW + jl;
    uint32_t acc = 0;
    if (j < n_words)
        for (uint32_t c = cl; c < n_chunks; c += CL) acc = kb::add(acc, partial[(size_t)c * n_words + j]);
    sm[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (uint32_t h = CL / 2; h >= 1; h >>= 1) {
        if (cl < h) sm[threadIdx.x] = kb::add(sm[threadIdx.x], sm[threadIdx.x + h * JW]);
        __syncthreads();
    }
    if (cl == 0 && j < n_words) out[j] = sm[jl];
}
static void launch_sum_partials(const uint32_t* partial, uint32_t n_chunks, uint32_t n_words, uint32_t* out, hipStream_t s) {
    if (n_words <= 4) hipLaunchKernelGGL(sum_partials_kernel<4>, dim3(1), dim3(256), 0, s, partial, n_chunks, n_words, out);
    else hipLaunchKernelGGL(sum_partials_kernel<64>, dim3((n_words + 63) / 64), dim3(256), 0, s, partial, n_chunks, n_words, out);
}

// partial[block] = sum_i eq[i] * mle[2 i] over the block's range (ext x ext)
__global__ __launch_bounds__(256) void fixed_at_zero_partial_kernel(const uint32_t* __restrict__ mle, uint32_t n,
                                                                    const uint32_t* __restrict__ eq,
                                                                    uint32_t* __restrict__ partial) {
    __shared__ uint32_t scratch[16];
    const uint32_t m = n >> 1;
    kb::Ext acc = kb::ext_zero();
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < m; i += gridDim.x * 256u) {
        kb::Ext a, e;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            a.c[k] = mle[(size_t)k * n + 2 * (size_t)i];
            e.c[k] = eq[(size_t)k * m + i];
        }
        acc = kb::ext_add(acc, kb::ext_mul(e, a));
    }
    block_sum4(acc.c, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) partial[blockIdx.x * 4 + k] = acc.c[k];
    }
}

// ---------------------------------------------------------------- fold-round leaves and openings
// Leaf i of a fold round = (cw[2i], cw[2i+1]) = 8 words = exactly one absorb block.
__global__ __launch_bounds__(256) void leaf_hash_pairs_kernel(const uint32_t* __restrict__ cw, uint32_t n,
                                                              const p2::RoundConstants* __restrict__ rc,
                                                              uint32_t* __restrict__ leaves) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (n >> 1)) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint2 v = *reinterpret_cast<const uint2*>(cw + (size_t)k * n + 2 * (size_t)i);
        s[k] = v.x;
        s[4 + k] = v.y;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) s[k] = 0;
    p2::permute(s, *rc);
    uint4* d = reinterpret_cast<uint4*>(leaves + (size_t)i * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

__global__ void open_pairs_kernel(const uint32_t* __restrict__ cw, uint32_t n, const uint32_t* __restrict__ indices,
                                  size_t n_idx, uint32_t* __restrict__ values) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * 8) return;
    const size_t q = t >> 3;
    const uint32_t w = (uint32_t)(t & 7);
    values[t] = cw[(size_t)(w & 3) * n + 2 * (size_t)indices[q] + (w >> 2)];
}

// Every fold round's opening in ONE launch (the query phase was 3 launches per round: shift the indices, gather the
// pairs, gather the paths): blockIdx.y = round r, whose index is q >> (r + 1); items [0, 8 n_idx) are the opened pair's
// words, the rest the path's half-digests (layer k of a tree of height h starts at digest 2^(h+1) - 2^(h-k+1)).
__global__ __launch_bounds__(256) void open_fold_rounds_kernel(const FoldOpenDesc* __restrict__ descs, const uint32_t* __restrict__ indices,
                                                               uint32_t n_idx, uint32_t* __restrict__ out) {
    const FoldOpenDesc d = descs[blockIdx.y];
    const uint32_t lg_h = d.lg_c - 1, r = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_idx * 8u) {
        const uint32_t q = t >> 3, w = t & 7u;
        const uint32_t idx = indices[q] >> (r + 1);
        out[d.vals_off + t] = d.cw[((size_t)(w & 3u) << d.lg_c) + 2 * (size_t)idx + (w >> 2)];
        return;
    }
    const uint32_t u = t - n_idx * 8u;
    if (u >= n_idx * lg_h * 2u) return;
    const uint32_t half = u & 1u, qk = u >> 1, k = qk % lg_h, q = qk / lg_h;
    const uint32_t idx = indices[q] >> (r + 1);
    const uint64_t off = ((uint64_t)2 << lg_h) - ((uint64_t)2 << (lg_h - k));
    const uint64_t node = off + ((idx >> k) ^ 1u);
    reinterpret_cast<uint4*>(out + d.paths_off + (size_t)qk * 8)[half] = reinterpret_cast<const uint4*>(d.tree + node * 8)[half];
}

__global__ void shift_indices_kernel(uint32_t* idx, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) idx[t] >>= 1;
}

}  // namespace sp1hip

using namespace sp1hip;

namespace sp1hip {
// internal entry points shared with prover.hip
int merkle_finish_tree(uint32_t* d_tree, int lg_height, uint32_t total_width, uint32_t* d_root_and_commit,
                       const DeviceCtx* ctx, hipStream_t s, const uint32_t* d_publish_extra = nullptr,
                       uint32_t* h_publish_slot = nullptr, uint32_t publish_seq = 0);

// h_publish_slot != null: the tree's last kernel also publishes [d_publish_extra[0..4) | root | commitment] to that mailbox
// slot with sequence number publish_seq (Mailbox::wait_next on the host side)
int commit_ext_pairs(const uint32_t* d_cw, int lg_n, uint32_t* d_tree, uint32_t* d_root_and_commit, hipStream_t s,
                     const uint32_t* d_publish_extra, uint32_t* h_publish_slot, uint32_t publish_seq) {
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    const uint32_t n = 1u << lg_n, leaves = n >> 1;
    hipLaunchKernelGGL(leaf_hash_pairs_kernel, dim3((leaves + 255) / 256), dim3(256), 0, s, d_cw, n, ctx->d_rc, d_tree);
    SP1HIP_LAUNCH_CHECK();
    return merkle_finish_tree(d_tree, lg_n - 1, 8, d_root_and_commit, ctx, s, d_publish_extra, h_publish_slot, publish_seq);
}

int open_ext_pairs(const uint32_t* d_cw, int lg_n, const uint32_t* d_indices, size_t n_idx, uint32_t* d_values,
                   hipStream_t s) {
    if (!n_idx) return SP1HIP_SUCCESS;
    hipLaunchKernelGGL(open_pairs_kernel, dim3((n_idx * 8 + 255) / 256), dim3(256), 0, s, d_cw, 1u << lg_n, d_indices,
                       n_idx, d_values);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int open_fold_rounds(const FoldOpenDesc* d_descs, int n_rounds, int max_lg_c, const uint32_t* d_indices, size_t n_idx,
                     uint32_t* d_out, hipStream_t s) {
    if (!n_idx || n_rounds <= 0) return SP1HIP_SUCCESS;
    const size_t items = n_idx * 8 + n_idx * (size_t)(max_lg_c - 1) * 2;
    hipLaunchKernelGGL(open_fold_rounds_kernel, dim3((unsigned)((items + 255) / 256), (unsigned)n_rounds), dim3(256), 0, s, d_descs,
                       d_indices, (uint32_t)n_idx, d_out);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int shift_indices(uint32_t* d_idx, size_t n, hipStream_t s) {
    if (!n) return SP1HIP_SUCCESS;
    hipLaunchKernelGGL(shift_indices_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_idx, n);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int ext_fixed_at_zero_async(const uint32_t* d_mle, int lg_n, const uint32_t* d_eq, uint32_t* d_out, hipStream_t s) {
    const uint32_t n = 1u << lg_n, m = n >> 1;
    uint32_t blocks = (m + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks == 0) blocks = 1;
    AsyncScratch part;
    SP1HIP_TRY(part.alloc((size_t)blocks * 16, s));
    hipLaunchKernelGGL(fixed_at_zero_partial_kernel, dim3(blocks), dim3(256), 0, s, d_mle, n, d_eq, (uint32_t*)part.p);
    SP1HIP_LAUNCH_CHECK();
    launch_sum_partials((const uint32_t*)part.p, blocks, 4u, d_out, s);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

// fold_round_kernel + the sum of its partials into d_zero_val (4 words); d_eq_next may be null (then d_zero_val is untouched)
// d_partial: scratch of 16 bytes per 256 entries of the folded message (caller-owned: no allocation between a round's launches)
int fold_round_async(const uint32_t* d_cw, int lg_c, const uint32_t* d_mle, int lg_m, const kb::Ext& beta, uint32_t* d_cw_out,
                     uint32_t* d_mle_out, const uint32_t* d_eq_next, uint32_t* d_zero_val, uint32_t* d_partial, hipStream_t s) {
    SP1HIP_REQUIRE(lg_c >= 1 && lg_c <= kb::TWO_ADICITY && lg_m >= 1 && lg_m <= 30, "fold sizes out of range");
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    ExtArg hb, b;
    for (int k = 0; k < 4; k++) { hb.c[k] = kb::mul(beta.c[k], 0x00ffffffu); b.c[k] = beta.c[k]; }
    const uint32_t cw_blocks = ((1u << (lg_c - 1)) + 255) / 256, mle_blocks = ((1u << (lg_m - 1)) + 255) / 256;
    hipLaunchKernelGGL(fold_round_kernel, dim3(cw_blocks + mle_blocks), dim3(256), 0, s, d_cw, lg_c, hb, ctx->d_tw_lo, ctx->d_tw_hi,
                       d_cw_out, cw_blocks, d_mle, lg_m, b, d_mle_out, d_eq_next, d_partial);
    SP1HIP_LAUNCH_CHECK();
    if (d_eq_next) {
        launch_sum_partials(d_partial, mle_blocks, 4u, d_zero_val, s);
        SP1HIP_LAUNCH_CHECK();
    }
    return SP1HIP_SUCCESS;
}

// d_out: 4 (2^(d+1) - 1) words; see eq_prefix_soa_kernel
int eq_prefix_tables_soa_async(const kb::Ext* h_point, int d, uint32_t* d_out, hipStream_t s) {
    SP1HIP_REQUIRE(d >= 0 && d <= kb::TWO_ADICITY + 6, "dim out of range");
    PointArg pt;
    for (int j = 0; j < d; j++)
        for (int k = 0; k < 4; k++) pt.c[j][k] = h_point[j].c[k];
    hipLaunchKernelGGL(eq_prefix_soa_kernel, dim3(((1u << d) + 255) / 256), dim3(256), 0, s, pt, d, d_out);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}
}  // namespace sp1hip

extern "C" {

int sp1hip_transpose_to_col_major(uint32_t* d_out, const uint32_t* d_in, size_t rows, size_t cols, sp1hip_stream_t stream) {
    if (rows == 0 || cols == 0) return SP1HIP_SUCCESS;
    SP1HIP_REQUIRE(d_out && d_in && d_out != d_in, "bad buffers");
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    SP1HIP_REQUIRE(grid.y <= 65535u * 1024u, "too many rows");
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, S(stream), d_in, d_out, rows, cols);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int sp1hip_transpose_to_row_major(uint32_t* d_out, const uint32_t* d_in, size_t rows, size_t cols, sp1hip_stream_t stream) {
    // column-major [rows x cols] is a row-major [cols][rows] array
    if (rows == 0 || cols == 0) return SP1HIP_SUCCESS;
    SP1HIP_REQUIRE(d_out && d_in && d_out != d_in, "bad buffers");
    dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, S(stream), d_in, d_out, cols, rows);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

static int monty_convert(uint32_t* d, size_t n, bool to, sp1hip_stream_t stream) {
    if (!n) return SP1HIP_SUCCESS;
    SP1HIP_REQUIRE(d, "null buffer");
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (to) hipLaunchKernelGGL(monty_convert_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, S(stream), d, n);
    else hipLaunchKernelGGL(monty_convert_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, S(stream), d, n);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}
int sp1hip_to_monty(uint32_t* d, size_t n, sp1hip_stream_t s) { return monty_convert(d, n, true, s); }
int sp1hip_from_monty(uint32_t* d, size_t n, sp1hip_stream_t s) { return monty_convert(d, n, false, s); }

int sp1hip_basefold_batch(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, const uint32_t* d_coeffs,
                          uint32_t* d_out, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_height >= 0 && lg_height <= 30, "lg_height out of range");
    SP1HIP_REQUIRE(d_coeffs && d_out, "null buffer");
    TensorTable tab;
    uint32_t tw;
    SP1HIP_TRY(make_tensor_table(tensors, n_tensors, &tab, &tw));
    SP1HIP_REQUIRE(tw <= 65536, "more than 2^16 columns in one message (batch_kernel's unreduced accumulators hold 2^16 terms)");
    hipStream_t s = S(stream);
    const uint32_t height = 1u << lg_height;
    AsyncScratch cols;
    SP1HIP_TRY(cols.alloc((size_t)tw * sizeof(uint32_t*), s));
    SP1HIP_TRY(expand_columns_async(tab, tw, height, (const uint32_t**)cols.p, s));
    hipLaunchKernelGGL(batch_kernel, dim3((height + 255) / 256), dim3(256), 0, s, (const uint32_t* const*)cols.p, tw,
                       height, d_coeffs, d_out);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int sp1hip_fold_even_odd(const uint32_t* d_cw, int lg_n, sp1hip_ext_t beta, uint32_t* d_out, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_n >= 1 && lg_n <= kb::TWO_ADICITY, "lg_n out of range");
    SP1HIP_REQUIRE(d_cw && d_out, "null buffer");
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    ExtArg hb;
    for (int k = 0; k < 4; k++) hb.c[k] = kb::mul(beta.c[k], 0x00ffffffu);  // beta / 2
    const uint32_t m = 1u << (lg_n - 1);
    hipLaunchKernelGGL(fold_even_odd_kernel, dim3((m + 255) / 256), dim3(256), 0, S(stream), d_cw, lg_n, hb,
                       ctx->d_tw_lo, ctx->d_tw_hi, d_out);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int sp1hip_fold_mle(const uint32_t* d_mle, int lg_n, sp1hip_ext_t beta, uint32_t* d_out, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_n >= 1 && lg_n <= 30, "lg_n out of range");
    SP1HIP_REQUIRE(d_mle && d_out, "null buffer");
    ExtArg b;
    for (int k = 0; k < 4; k++) b.c[k] = beta.c[k];
    const uint32_t m = 1u << (lg_n - 1);
    hipLaunchKernelGGL(fold_mle_kernel, dim3((m + 255) / 256), dim3(256), 0, S(stream), d_mle, lg_n, b, d_out);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int sp1hip_partial_lagrange(const sp1hip_ext_t* h_point, int dim, uint32_t* d_out, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(dim >= 0 && dim <= kb::TWO_ADICITY + 6, "dim out of range");
    SP1HIP_REQUIRE(d_out && (h_point || dim == 0), "null buffer");
    hipStream_t s = S(stream);
    PointArg pt;
    for (int j = 0; j < dim; j++)
        for (int k = 0; k < 4; k++) pt.c[j][k] = h_point[j].c[k];
    const int d_hi = dim / 2, d_lo = dim - d_hi;
    AsyncScratch small;
    SP1HIP_TRY(small.alloc((((size_t)1 << d_hi) + ((size_t)1 << d_lo)) * 16, s));
    uint32_t* hi = (uint32_t*)small.p;
    uint32_t* lo = hi + ((size_t)4 << d_hi);
    hipLaunchKernelGGL(eq_small_kernel, dim3(((1u << d_hi) + 255) / 256), dim3(256), 0, s, pt, 0, d_hi, hi);
    SP1HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(eq_small_kernel, dim3(((1u << d_lo) + 255) / 256), dim3(256), 0, s, pt, d_hi, d_lo, lo);
    SP1HIP_LAUNCH_CHECK();
    const uint32_t len = 1u << dim;
    hipLaunchKernelGGL(eq_outer_kernel, dim3((len + 255) / 256), dim3(256), 0, s, hi, lo, d_lo, len, d_out);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int sp1hip_mle_eval_columns(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, const uint32_t* d_eq,
                            uint32_t* d_evals, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_height >= 0 && lg_height <= 30, "lg_height out of range");
    SP1HIP_REQUIRE(d_eq && d_evals, "null buffer");
    TensorTable tab;
    uint32_t tw;
    SP1HIP_TRY(make_tensor_table(tensors, n_tensors, &tab, &tw));
    if (tw == 0) return SP1HIP_SUCCESS;
    SP1HIP_REQUIRE(tw <= 65536, "more than 2^16 columns in one message (the unreduced accumulators hold 2^16 terms)");
    hipStream_t s = S(stream);
    const uint32_t height = 1u << lg_height;
    AsyncScratch cols, part;
    SP1HIP_TRY(cols.alloc((size_t)tw * sizeof(uint32_t*), s));
    SP1HIP_TRY(expand_columns_async(tab, tw, height, (const uint32_t**)cols.p, s));
    const uint32_t chunks = (height + EVAL_ROWS - 1) / EVAL_ROWS;
    SP1HIP_TRY(part.alloc((size_t)chunks * tw * 16, s));
    dim3 grid((tw + EVAL_COLS - 1) / EVAL_COLS, chunks);
    hipLaunchKernelGGL(eval_columns_partial_kernel, grid, dim3(256), 0, s, (const uint32_t* const*)cols.p, tw, height,
                       d_eq, (uint32_t*)part.p);
    SP1HIP_LAUNCH_CHECK();
    launch_sum_partials((const uint32_t*)part.p, chunks, tw * 4, d_evals, s);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int sp1hip_ext_fixed_at_zero(const uint32_t* d_mle, int lg_n, const uint32_t* d_eq, uint32_t* d_out,
                             sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_n >= 1 && lg_n <= 30, "lg_n out of range");
    SP1HIP_REQUIRE(d_mle && d_eq && d_out, "null buffer");
    return ext_fixed_at_zero_async(d_mle, lg_n, d_eq, d_out, S(stream));
}

}  // extern "C"
