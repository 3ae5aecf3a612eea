// sp1_amd/csrc/merkle.hip — Poseidon2 Merkle tensor commitment on gfx950.
//
// Replaces `FieldMerkleTreeProver::{commit_tensors, prove_openings_at_indices}` and
// `compute_openings_at_indices` (/root/reference/slop/crates/merkle-tree/src/p3sync.rs:L40-L238) for
// `Poseidon2KoalaBear16Prover`; the commitment finalisation compress(root, hash([log_height, width]))
// is p3sync.rs:L136-L142.
//
// Kernel shapes:
//  * leaf_hash: one lane per row. A lane walks the concatenated row 8 columns at a time; every
//    column load is coalesced across the wave (consecutive rows of one column-major column), the
//    16-word sponge state stays in VGPRs across all absorb blocks. ALU-bound (~12k integer op
//    slots per permutation vs 32 bytes loaded), so no LDS staging: extra waves, not tiling, hide
//    the load latency.
//  * compress_layer: one lane per parent, 64 B in (2 x dwordx4 x 2), 32 B out.
//  * compress_layer_coop: layers of <= 8192 parents, 16 lanes per compression (latency, not throughput, is what a small
//    layer costs); compress_top: the last <= 7 levels in one workgroup, cooperative form throughout.
#include <algorithm>
#include <cstdlib>

#include "device_ctx.hpp"
#include "tensor_table.hpp"

namespace sp1hip {

__global__ void expand_columns_kernel(TensorTable tab, uint32_t total_width, uint64_t height, const uint32_t** out) {
    for (uint32_t g = threadIdx.x; g < total_width; g += blockDim.x) {
        int t = 0;
        while (tab.col_start[t + 1] <= g) t++;
        out[g] = tab.base[t] + (uint64_t)(g - tab.col_start[t]) * height;
    }
}

int make_tensor_table(const sp1hip_tensor_t* tensors, int n_tensors, TensorTable* tab, uint32_t* total_width) {
    SP1HIP_REQUIRE(tensors && n_tensors > 0, "empty tensor message");
    SP1HIP_REQUIRE(n_tensors <= MAX_TENSORS, "too many tensors in one message (max 256)");
    uint32_t w = 0;
    for (int i = 0; i < n_tensors; i++) {
        SP1HIP_REQUIRE(tensors[i].d_data != nullptr || tensors[i].width == 0, "null tensor data");
        tab->base[i] = tensors[i].d_data;
        tab->col_start[i] = w;
        w += tensors[i].width;
    }
    tab->col_start[n_tensors] = w;
    tab->n = n_tensors;
    *total_width = w;
    return SP1HIP_SUCCESS;
}

int expand_columns_async(const TensorTable& tab, uint32_t total_width, uint64_t height, const uint32_t** d_cols,
                         hipStream_t stream) {
    if (total_width == 0) return SP1HIP_SUCCESS;
    hipLaunchKernelGGL(expand_columns_kernel, dim3(1), dim3(256), 0, stream, tab, total_width, height, d_cols);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

__device__ __forceinline__ void store_digest(uint32_t* dst, const uint32_t (&s)[16]) {
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// The sponge state lives in VGPRs as doubles across all absorb blocks of a row (poseidon2.hpp permute_f64): the
// capacity lanes are never reduced between permutations and the rate lanes a block overwrites are never reduced at
// all; only the 8 digest words are brought back to canonical form at the end.
__global__ __launch_bounds__(256) void leaf_hash_kernel(const uint32_t* const* __restrict__ cols, uint32_t total_width,
                                                        uint32_t height, const p2::RoundConstants* __restrict__ rc,
                                                        uint32_t* __restrict__ leaves) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= height) return;
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = 0.0;
    const uint32_t full = total_width >> 3;
    for (uint32_t k = 0; k < full; k++) {
#pragma unroll
        for (int j = 0; j < 8; j++) d[j] = (double)gptr(cols[8 * k + j])[row];
        p2::permute_f64(d, *rc);
    }
    const uint32_t rem = total_width & 7u;
    if (rem) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < rem) d[j] = (double)gptr(cols[8 * full + j])[row];
        p2::permute_f64(d, *rc);
    }
    uint32_t s[16];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = p2::canonical_f64(d[j]);
    store_digest(leaves + (size_t)row * 8, s);
}

// The same sponge split at tensor boundaries: part k absorbs columns [c0, c0 + width) of the concatenated row.
// Valid when every boundary falls on a multiple of the rate (8) and every part but the first starts with a
// full block, so only the 8 capacity words cross a boundary (`carry`, column-major [8][height]): the rate
// words are overwritten by the next block anyway (PaddingFreeSponge overwrite mode). Lets commit_mles hash
// tensor k while tensor k + 1 is still being encoded on another stream.
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void leaf_hash_part_kernel(const uint32_t* const* __restrict__ cols, uint32_t width,
                                                             uint32_t height, const p2::RoundConstants* __restrict__ rc,
                                                             uint32_t* __restrict__ carry, uint32_t* __restrict__ leaves) {
    // grid-stride over the rows: the launch may be a PERSISTENT grid (leaf_hash_part: a bounded number of workgroups per
    // CU, so that the encode passes of the next batch, running on the side stream, always find wave slots next to it)
    for (uint32_t row = blockIdx.x * 256u + threadIdx.x; row < height; row += gridDim.x * 256u) {
        double d[16];
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; i++) d[8 + i] = FIRST ? 0.0 : (double)carry[(size_t)i * height + row];
        const uint32_t full = width >> 3;
        for (uint32_t k = 0; k < full; k++) {
#pragma unroll
            for (int j = 0; j < 8; j++) d[j] = (double)gptr(cols[8 * k + j])[row];
            p2::permute_f64(d, *rc);
        }
        if (LAST) {
            const uint32_t rem = width & 7u;
            if (rem) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if ((uint32_t)j < rem) d[j] = (double)gptr(cols[8 * full + j])[row];
                p2::permute_f64(d, *rc);
            }
            uint32_t s[16];
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] = p2::canonical_f64(d[j]);
            store_digest(leaves + (size_t)row * 8, s);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) carry[(size_t)i * height + row] = p2::canonical_f64(d[8 + i]);
        }
    }
}

// Groups consecutive tensors into parts that leaf_hash_part_kernel can hash one after the other: a part is
// closed as soon as the concatenated row so far is a whole number of blocks; a trailing remainder narrower
// than one block joins the part before it. parts[k] = {last tensor of part k, first column, width}.
// Fewer than two parts: hash in one launch instead.
void leaf_hash_plan(const sp1hip_tensor_t* tensors, int n_tensors, std::vector<LeafPart>* parts) {
    parts->clear();
    uint32_t c0 = 0, c = 0;
    for (int i = 0; i < n_tensors; i++) {
        c += tensors[i].width;
        if (c > c0 && ((c - c0) & 7u) == 0) {
            parts->push_back({i, c0, c - c0});
            c0 = c;
        }
    }
    if (c > c0) {
        if (c - c0 >= 8 || parts->empty()) parts->push_back({n_tensors - 1, c0, c - c0});
        else { parts->back().last_tensor = n_tensors - 1; parts->back().width += c - c0; }
    } else if (!parts->empty()) {
        parts->back().last_tensor = n_tensors - 1;      // zero-width tensors at the end
    }
}

// Enqueues part `k` of `n_parts`: d_cols points at the column table entry of the part's first column.
int leaf_hash_part(const uint32_t* const* d_cols, uint32_t width, int k, int n_parts, uint32_t height, uint32_t* d_carry,
                   uint32_t* d_tree, const DeviceCtx* ctx, hipStream_t s) {
    ScopedTimer t("leaf_hash", s);
    // SP1HIP_LEAF_WGS=n: at most n workgroups (persistent grid); 0 / unset: one workgroup per 256 rows
    const uint32_t cap = [] { const char* e = getenv("SP1HIP_LEAF_WGS"); return e ? (uint32_t)atoi(e) : 0u; }();
    const uint32_t wgs = (height + 255) / 256;
    const dim3 grid(cap ? std::min(cap, wgs) : wgs), block(256);
    const bool first = k == 0, last = k == n_parts - 1;
    if (first && last) return SP1HIP_ERROR_INVALID_ARGUMENT;
    if (first)
        hipLaunchKernelGGL((leaf_hash_part_kernel<true, false>), grid, block, 0, s, d_cols, width, height, ctx->d_rc, d_carry, d_tree);
    else if (last)
        hipLaunchKernelGGL((leaf_hash_part_kernel<false, true>), grid, block, 0, s, d_cols, width, height, ctx->d_rc, d_carry, d_tree);
    else
        hipLaunchKernelGGL((leaf_hash_part_kernel<false, false>), grid, block, 0, s, d_cols, width, height, ctx->d_rc, d_carry, d_tree);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

__device__ __forceinline__ void load_pair(const uint32_t* src, uint32_t (&s)[16]) {
    const uint4* p = reinterpret_cast<const uint4*>(src);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
    s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    s[8] = c.x; s[9] = c.y; s[10] = c.z; s[11] = c.w;
    s[12] = d.x; s[13] = d.y; s[14] = d.z; s[15] = d.w;
}

__global__ __launch_bounds__(256) void compress_layer_kernel(const uint32_t* __restrict__ children, uint32_t n_parents,
                                                             const p2::RoundConstants* __restrict__ rc,
                                                             uint32_t* __restrict__ parents) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_parents) return;
    uint32_t s[16];
    load_pair(children + (size_t)i * 16, s);
    p2::permute(s, *rc);
    store_digest(parents + (size_t)i * 8, s);
}

// The same layer in the cooperative form: 16 lanes per compression (lane r holds word r of the two children, which are 16
// consecutive words), linear layers as DPP shuffles (poseidon2.hpp). It does ~6x the VALU work of the per-lane form, so it
// is for the SMALL layers only (<= COOP_LAYER_MAX parents, where the chip is nearly empty and a layer's time is the latency
// of ONE compression): that latency is ~2.3x shorter here, and a Merkle tree's tail is a chain of such layers — 21 BaseFold
// trees and 7 commit trees per shard proof. n_parents is a multiple of 4 (whole waves: DPP rows must be fully active).
__global__ __launch_bounds__(256) void compress_layer_coop_kernel(const uint32_t* __restrict__ children, uint32_t n_parents,
                                                                  const p2::RoundConstants* __restrict__ rc,
                                                                  uint32_t* __restrict__ parents) {
    const uint32_t lane = threadIdx.x & 15u, row = blockIdx.x * 16u + (threadIdx.x >> 4);
    const uint32_t wave_row0 = blockIdx.x * 16u + ((threadIdx.x >> 6) << 2);
    if (wave_row0 >= n_parents) return;                                   // wave-uniform
    uint32_t x = children[(size_t)row * 16 + lane];
    x = p2::permute_coop16(x, lane, *rc);
    if (lane < 8) parents[(size_t)row * 8 + lane] = x;
}
constexpr uint32_t COOP_LAYER_MAX = 8192, TOP_MAX = 128;

// One workgroup finishes the tree: `layer` holds n (<= 2048, power of two; the host hands over <= TOP_MAX) digests and the parents are laid out
// right behind it, level after level. Also writes root and the finalised commitment. Levels with >= 64 parents
// use one lane per compression; the last levels (a chain of dependent compressions with almost no parallelism,
// i.e. pure latency) use the cooperative permutation — 16 lanes per compression, DPP shuffles for the linear
// layers (poseidon2.hpp) — in as few waves as the level needs, so a wave issues back to back (measured: one
// cooperative compression is ~2.3x shorter than a per-lane one when the SIMD is not shared).
__global__ __launch_bounds__(1024) void compress_top_kernel(uint32_t* layer, uint32_t n, uint32_t lg_height,
                                                            uint32_t total_width,
                                                            const p2::RoundConstants* __restrict__ rc,
                                                            uint32_t* __restrict__ root_and_commit,
                                                            const uint32_t* __restrict__ publish_extra,
                                                            volatile uint32_t* publish_slot, uint32_t publish_seq) {
    const uint32_t lane = threadIdx.x & 15u, row = threadIdx.x >> 4;      // cooperative view: 64 rows of 16 lanes
    const uint32_t wave_row0 = (threadIdx.x >> 6) << 2;                   // first row of this wave (wave-uniform)
    uint32_t* cur = layer;
    while (n > 1) {
        uint32_t* nxt = cur + (size_t)n * 8;
        const uint32_t np = n >> 1;
        if (np > 64) {
            for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
                uint32_t s[16];
                load_pair(cur + (size_t)i * 16, s);
                p2::permute(s, *rc);
                store_digest(nxt + (size_t)i * 8, s);
            }
        } else if (wave_row0 < np) {                                      // whole waves skip: DPP needs full rows only
            const bool valid = row < np;
            uint32_t x = valid ? cur[(size_t)row * 16 + lane] : 0u;       // the two children are 16 consecutive words
            x = p2::permute_coop16(x, lane, *rc);
            if (valid && lane < 8) nxt[(size_t)row * 8 + lane] = x;
        }
        __syncthreads();
        cur = nxt;
        n = np;
    }
    if (threadIdx.x >= 64) return;
    // commitment = compress(root, hash([lg_height, total_width])): wave 0, every row computes it, row 0 stores it
    uint32_t h = lane == 0 ? kb::to_monty(lg_height) : (lane == 1 ? kb::to_monty(total_width) : 0u);
    h = p2::permute_coop16(h, lane, *rc);
    const uint32_t h_up = p2::dpp_mov<p2::DPP_ROW_ROR8>(h);              // digest words 0..7 moved to lanes 8..15
    const uint32_t root_word = cur[lane & 7u];
    uint32_t x = lane < 8 ? root_word : h_up;
    x = p2::permute_coop16(x, lane, *rc);
    if (row == 0 && lane < 8) {
        root_and_commit[lane] = root_word;
        root_and_commit[8 + lane] = x;
    }
    // Optional hand-over to the host in the same launch (a BaseFold round: the prover waits for [4 extra words | root |
    // commitment]; a mailbox kernel behind this one was one more launch per round): payload, then — once the stores are
    // acknowledged — the sequence number, like mailbox_publish_kernel (runtime.hip).
    if (publish_slot == nullptr) return;
    if (row == 0) {
        if (lane < 4) publish_slot[1 + lane] = publish_extra[lane];
        if (lane < 8) {
            publish_slot[5 + lane] = root_word;
            publish_slot[13 + lane] = x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (only wave 0 is left here: its payload stores are acknowledged — vmcnt counts the whole wave — then ONE release store)
    if (threadIdx.x == 0) __hip_atomic_store(const_cast<uint32_t*>(publish_slot), publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool INTEGER_FORM>
__global__ void permute_states_kernel(uint32_t* states, size_t n, const p2::RoundConstants* __restrict__ rc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = states[i * 16 + k];
    if (INTEGER_FORM) p2::permute_int(s, *rc); else p2::permute(s, *rc);
#pragma unroll
    for (int k = 0; k < 16; k++) states[i * 16 + k] = s[k];
}

__global__ void open_values_kernel(const uint32_t* const* __restrict__ cols, uint32_t total_width,
                                   const uint32_t* __restrict__ indices, size_t n_idx, uint32_t* __restrict__ values) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * total_width) return;
    size_t q = t / total_width;
    uint32_t g = (uint32_t)(t % total_width);
    values[t] = gptr(cols[g])[indices[q]];
}

// paths[q][k][0..8] = layer_k[(idx >> k) ^ 1]; one lane per (q, k, half-digest)
__global__ void open_paths_kernel(const uint32_t* __restrict__ tree, uint32_t lg_height,
                                  const uint32_t* __restrict__ indices, size_t n_idx, uint32_t* __restrict__ paths) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * lg_height * 2) return;
    uint32_t half = (uint32_t)(t & 1);
    size_t qk = t >> 1;
    uint32_t k = (uint32_t)(qk % lg_height);
    size_t q = qk / lg_height;
    // layer k starts at digest offset 2^(h+1) - 2^(h-k+1)
    uint64_t off = ((uint64_t)2 << lg_height) - ((uint64_t)2 << (lg_height - k));
    uint64_t node = off + ((indices[q] >> k) ^ 1u);
    const uint4* src = reinterpret_cast<const uint4*>(tree + node * 8) + half;
    reinterpret_cast<uint4*>(paths + (qk * 8))[half] = *src;
}

// Compresses the leaf layer at d_tree up to the root and finalises the commitment.
int merkle_finish_tree(uint32_t* d_tree, int lg_height, uint32_t total_width, uint32_t* d_root_and_commit,
                       const DeviceCtx* ctx, hipStream_t s, const uint32_t* d_publish_extra, uint32_t* h_publish_slot,
                       uint32_t publish_seq) {
    ScopedTimer t("compress", s);
    uint32_t* cur = d_tree;
    uint32_t n = 1u << lg_height;
    while (n > TOP_MAX) {
        uint32_t* nxt = cur + (size_t)n * 8;
        const uint32_t np = n / 2;
        if (np > COOP_LAYER_MAX)
            hipLaunchKernelGGL(compress_layer_kernel, dim3((np + 255) / 256), dim3(256), 0, s, cur, np, ctx->d_rc, nxt);
        else
            hipLaunchKernelGGL(compress_layer_coop_kernel, dim3((np + 15) / 16), dim3(256), 0, s, cur, np, ctx->d_rc, nxt);
        SP1HIP_LAUNCH_CHECK();
        cur = nxt;
        n >>= 1;
    }
    hipLaunchKernelGGL(compress_top_kernel, dim3(1), dim3(1024), 0, s, cur, n, (uint32_t)lg_height, total_width,
                       ctx->d_rc, d_root_and_commit, d_publish_extra, (volatile uint32_t*)h_publish_slot, publish_seq);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_merkle_commit(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, uint32_t* d_tree,
                         uint32_t* d_root_and_commit, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_height >= 0 && lg_height <= 30, "lg_height out of range");
    SP1HIP_REQUIRE(d_tree && d_root_and_commit, "null output");
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    TensorTable tab;
    uint32_t tw;
    SP1HIP_TRY(make_tensor_table(tensors, n_tensors, &tab, &tw));
    const uint32_t height = 1u << lg_height;
    hipStream_t s = S(stream);
    AsyncScratch cols;
    SP1HIP_TRY(cols.alloc((size_t)tw * sizeof(uint32_t*), s));
    SP1HIP_TRY(expand_columns_async(tab, tw, height, (const uint32_t**)cols.p, s));
    {
        ScopedTimer t("leaf_hash", s);
        hipLaunchKernelGGL(leaf_hash_kernel, dim3((height + 255) / 256), dim3(256), 0, s,
                           (const uint32_t* const*)cols.p, tw, height, ctx->d_rc, d_tree);
    }
    SP1HIP_LAUNCH_CHECK();
    return merkle_finish_tree(d_tree, lg_height, tw, d_root_and_commit, ctx, s, nullptr, nullptr, 0);
}

int sp1hip_merkle_open(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, const uint32_t* d_tree,
                       const uint32_t* d_indices, size_t n_idx, uint32_t* d_values, uint32_t* d_paths,
                       sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(lg_height >= 0 && lg_height <= 30, "lg_height out of range");
    SP1HIP_REQUIRE(d_indices || n_idx == 0, "null indices");
    if (n_idx == 0) return SP1HIP_SUCCESS;
    hipStream_t s = S(stream);
    if (d_values) {
        TensorTable tab;
        uint32_t tw;
        SP1HIP_TRY(make_tensor_table(tensors, n_tensors, &tab, &tw));
        AsyncScratch cols;
        SP1HIP_TRY(cols.alloc((size_t)tw * sizeof(uint32_t*), s));
        SP1HIP_TRY(expand_columns_async(tab, tw, (uint64_t)1 << lg_height, (const uint32_t**)cols.p, s));
        size_t total = n_idx * tw;
        if (total) {
            hipLaunchKernelGGL(open_values_kernel, dim3((total + 255) / 256), dim3(256), 0, s,
                               (const uint32_t* const*)cols.p, tw, d_indices, n_idx, d_values);
            SP1HIP_LAUNCH_CHECK();
        }
    }
    if (d_paths && lg_height > 0) {
        SP1HIP_REQUIRE(d_tree, "null tree");
        size_t total = n_idx * lg_height * 2;
        hipLaunchKernelGGL(open_paths_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d_tree, (uint32_t)lg_height,
                           d_indices, n_idx, d_paths);
        SP1HIP_LAUNCH_CHECK();
    }
    return SP1HIP_SUCCESS;
}

int sp1hip_poseidon2_permute(uint32_t* d_states, size_t n, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(d_states || n == 0, "null states");
    if (!n) return SP1HIP_SUCCESS;
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    hipLaunchKernelGGL(permute_states_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, S(stream), d_states, n, ctx->d_rc);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

int sp1hip_poseidon2_permute_integer_form(uint32_t* d_states, size_t n, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(d_states || n == 0, "null states");
    if (n == 0) return SP1HIP_SUCCESS;
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    hipLaunchKernelGGL(permute_states_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, S(stream), d_states, n, ctx->d_rc);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

}  // extern "C"
