// sp1_amd/csrc/stacked.hip — the commit wrappers above BaseFold: stacked PCS (a5) and jagged PCS (a7, a8).
//
//   sp1hip_stacked_commit   `StackedPcsProver::commit_multilinears`  /root/reference/slop/crates/stacked/src/prover.rs:L59-L94
//                           `interleave_multilinears_with_fixed_rate` /root/reference/slop/crates/stacked/src/fixed_rate.rs:L6-L47
//   sp1hip_jagged_commit    `JaggedProver::commit_multilinears`      /root/reference/slop/crates/jagged/src/prover.rs:L106-L160
//                           (as called by `ShardProver::commit_traces`, /root/reference/crates/hypercube/src/prover/shard.rs:L462-L468)
//
// MI355X shape: chip tables arrive column-major, and a column-major table IS the concatenation of its
// columns, so the reference's transpose + flatten + split_off + transpose dance collapses to one
// device-to-device copy per table into a dense buffer that is zero-padded to a multiple of the
// stacking height; the stacked batches `[2^lsh x batch]` handed to BaseFold are *slices* of that
// buffer (no interleave kernel, no second copy).
#include <cstring>
#include <memory>
#include <algorithm>
#include <functional>
#include <vector>

#include "device_ctx.hpp"
#include "round_sync.hpp"
#include "stacked_data.hpp"
#include "tensor_table.hpp"

using namespace sp1hip;

namespace sp1hip {     // prover.hip
int commit_mles_hooked(const sp1hip_tensor_t* mles, int n_mles, int lg_n, int lg_blowup, uint32_t h_commit[8],
                       sp1hip_basefold_data_t** out, sp1hip_stream_t stream, const std::function<int(int, hipStream_t)>* before_encode);
}

namespace {
// table slices -> their place in the dense (stacked) buffer: blockIdx.y = slice, the x blocks stride over its words
struct DenseSeg { const uint32_t* src; uint64_t dst_off, n; };
__global__ __launch_bounds__(256) void dense_fill_kernel(const DenseSeg* __restrict__ segs, uint32_t* __restrict__ dense) {
    const DenseSeg sg = segs[blockIdx.y];
    uint32_t* dst = dense + sg.dst_off;
    for (uint64_t base = (uint64_t)blockIdx.x * 1024u; base < sg.n; base += (uint64_t)gridDim.x * 1024u) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t i = base + (uint64_t)k * 256u + threadIdx.x;
            if (i < sg.n) dst[i] = sg.src[i];
        }
    }
}

// PaddingFreeSponge on the host (metadata hashes: a handful of permutations)
void host_hash(const std::vector<uint32_t>& in, uint32_t out[8]) {
    uint32_t s[16] = {0};
    size_t fill = 0;
    for (uint32_t x : in) {
        s[fill++] = x;
        if (fill == 8) { p2_host_permute(s); fill = 0; }
    }
    if (fill) p2_host_permute(s);
    memcpy(out, s, 32);
}
void host_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
    uint32_t s[16];
    memcpy(s, l, 32);
    memcpy(s + 8, r, 32);
    p2_host_permute(s);
    memcpy(out, s, 32);
}
}  // namespace

extern "C" {

int sp1hip_stacked_commit(const sp1hip_table_t* tables, int n_tables, int log_stacking_height, int batch_size,
                          int lg_blowup, uint32_t h_commit[8], uint64_t* num_added_vals, sp1hip_stacked_data_t** out,
                          sp1hip_stream_t stream) {
    SP1HIP_REQUIRE((tables || n_tables == 0) && n_tables >= 0 && h_commit && out, "bad argument");
    SP1HIP_REQUIRE(log_stacking_height >= 0 && log_stacking_height + lg_blowup <= kb::TWO_ADICITY, "stacking height out of range");
    SP1HIP_REQUIRE(batch_size >= 1, "batch_size must be positive");
    hipStream_t s = S(stream);
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    const uint64_t H = (uint64_t)1 << log_stacking_height;
    uint64_t area = 0;
    for (int i = 0; i < n_tables; i++) {
        SP1HIP_REQUIRE(tables[i].d_data || tables[i].rows * tables[i].cols == 0, "null table data");
        area += tables[i].rows * (uint64_t)tables[i].cols;
    }
    // next multiple of the stacking height, at least one column (prover.rs:L72-L79)
    uint64_t padded = ((area + H - 1) / H) * H;
    if (padded < H) padded = H;
    std::unique_ptr<sp1hip_stacked_data_s> sd(new sp1hip_stacked_data_s());
    sd->stream = s;
    sd->area = area;
    sd->padded = padded;
    sd->log_stacking_height = log_stacking_height;
    SP1HIP_TRY(arena_alloc(&sd->d_dense, padded * 4, s));
    // The tables are concatenated into the dense buffer batch by batch, each batch right before ITS encode and on the stream
    // that encodes it (commit_mles_hooked): with the encodes on the side stream the copies (1.6 GB of a core shard, 0.7 ms
    // of HBM time) run under the VALU-bound leaf hashes of the batches before instead of in front of the commitment.
    // One gather launch per batch (a hipMemcpyAsync per table slice was 35 API calls of ~30 us each at the head of a proof:
    // the host, not the copies, was what the first leaf hash waited for).
    std::vector<uint64_t> table_off(n_tables + 1, 0);
    for (int i = 0; i < n_tables; i++) table_off[i + 1] = table_off[i] + tables[i].rows * (uint64_t)tables[i].cols;
    if (padded > area) SP1HIP_HIP(hipMemsetAsync((uint32_t*)sd->d_dense + area, 0, (padded - area) * 4, s));
    uint32_t* const dense = (uint32_t*)sd->d_dense;
    const uint64_t batch_words = (uint64_t)batch_size * H;
    const uint64_t n_batches_total = area == 0 ? 1 : (padded / H + (uint64_t)batch_size - 1) / (uint64_t)batch_size;
    std::vector<DenseSeg> segs;
    std::vector<std::pair<uint32_t, uint32_t>> batch_segs(n_batches_total, {0u, 0u});     // (first segment, count)
    std::vector<uint64_t> batch_max(n_batches_total, 0);
    for (uint64_t b = 0; b < n_batches_total; b++) {
        const uint64_t lo = b * batch_words, hi = std::min<uint64_t>(lo + batch_words, area);
        batch_segs[b].first = (uint32_t)segs.size();
        for (int i = 0; i < n_tables && lo < hi; i++) {
            const uint64_t a = std::max(lo, table_off[i]), e = std::min(hi, table_off[i + 1]);
            if (a < e) { segs.push_back(DenseSeg{tables[i].d_data + (a - table_off[i]), a, e - a}); batch_max[b] = std::max(batch_max[b], e - a); }
        }
        batch_segs[b].second = (uint32_t)segs.size() - batch_segs[b].first;
    }
    AsyncScratch d_segs;
    PinnedStage seg_stage;
    SP1HIP_TRY(seg_stage.init(s));
    SP1HIP_TRY(d_segs.alloc(std::max<size_t>(segs.size(), 1) * sizeof(DenseSeg), s));
    SP1HIP_TRY(seg_stage.upload(d_segs.p, segs.data(), segs.size() * sizeof(DenseSeg)));
    const std::function<int(int, hipStream_t)> fill_batch = [&](int b, hipStream_t on) -> int {
        if ((uint64_t)b >= n_batches_total || batch_segs[b].second == 0) return SP1HIP_SUCCESS;
        const uint32_t gx = (uint32_t)std::min<uint64_t>((batch_max[b] + 1023) / 1024, 2048);
        hipLaunchKernelGGL(dense_fill_kernel, dim3(gx, batch_segs[b].second), dim3(256), 0, on,
                           (const DenseSeg*)d_segs.p + batch_segs[b].first, dense);
        SP1HIP_LAUNCH_CHECK();
        return SP1HIP_SUCCESS;
    };
    const uint64_t ncols = area == 0 ? 0 : padded / H;
    // an empty message yields ONE zero-width batch, as the reference's interleave does (fixed_rate.rs:L38-L44)
    if (ncols == 0) sd->batches.push_back({(const uint32_t*)sd->d_dense, 0u});
    for (uint64_t c0 = 0; c0 < ncols; c0 += (uint64_t)batch_size) {
        const uint32_t w = (uint32_t)std::min<uint64_t>((uint64_t)batch_size, ncols - c0);
        sd->batches.push_back({(const uint32_t*)sd->d_dense + c0 * H, w});
    }
    SP1HIP_REQUIRE(sd->batches.size() <= 128, "more than 128 stacked batches in one commitment");
    SP1HIP_TRY(commit_mles_hooked(sd->batches.data(), (int)sd->batches.size(), log_stacking_height, lg_blowup, sd->commit,
                                  &sd->basefold, stream, &fill_batch));
    memcpy(h_commit, sd->commit, 32);
    if (num_added_vals) *num_added_vals = padded - area;
    *out = sd.release();
    return SP1HIP_SUCCESS;
}

void sp1hip_stacked_data_free(sp1hip_stacked_data_t* data) { delete data; }

int sp1hip_stacked_data_info(const sp1hip_stacked_data_t* data, sp1hip_basefold_data_t** basefold, int* n_batches,
                             const uint32_t** d_dense, uint64_t* padded_area) {
    SP1HIP_REQUIRE(data, "null argument");
    if (basefold) *basefold = data->basefold;
    if (n_batches) *n_batches = (int)data->batches.size();
    if (d_dense) *d_dense = (const uint32_t*)data->d_dense;
    if (padded_area) *padded_area = data->padded;
    return SP1HIP_SUCCESS;
}

int sp1hip_stacked_batch(const sp1hip_stacked_data_t* data, int k, sp1hip_tensor_t* batch) {
    SP1HIP_REQUIRE(data && batch && k >= 0 && k < (int)data->batches.size(), "bad argument");
    *batch = data->batches[k];
    return SP1HIP_SUCCESS;
}

int sp1hip_jagged_commit(const sp1hip_table_t* tables, int n_tables, int max_log_row_count, int log_stacking_height,
                         int batch_size, int lg_blowup, uint32_t h_commit[8], sp1hip_stacked_data_t** out,
                         sp1hip_stream_t stream) {
    SP1HIP_REQUIRE((tables || n_tables == 0) && h_commit && out, "bad argument");
    SP1HIP_REQUIRE(max_log_row_count >= 0 && max_log_row_count <= 30, "max_log_row_count out of range");
    const uint64_t M = (uint64_t)1 << max_log_row_count;
    // only tables with real rows go to the dense PCS (prover.rs:L129-L131); all of them are counted
    std::vector<sp1hip_table_t> dense;
    std::vector<uint64_t> rows, cols;
    for (int i = 0; i < n_tables; i++) {
        SP1HIP_REQUIRE(tables[i].rows <= M, "table taller than 2^max_log_row_count");
        rows.push_back(tables[i].rows);
        cols.push_back(tables[i].cols);
        if (tables[i].rows) dense.push_back(tables[i]);
    }
    uint32_t inner[8];
    uint64_t added = 0;
    SP1HIP_TRY(sp1hip_stacked_commit(dense.data(), (int)dense.size(), log_stacking_height, batch_size, lg_blowup, inner,
                                     &added, out, stream));
    // two dummy tables account for the stacking padding (prover.rs:L133-L139)
    uint64_t added_cols = (added + M - 1) / M;
    if (added_cols < 1) added_cols = 1;
    rows.push_back(M);
    rows.push_back(added - (added_cols - 1) * M);
    cols.push_back(added_cols - 1);
    cols.push_back(1);
    std::vector<uint32_t> meta;
    meta.push_back(kb::to_monty((uint32_t)rows.size()));
    for (uint64_t r : rows) meta.push_back(kb::to_monty((uint32_t)r));
    for (uint64_t c : cols) meta.push_back(kb::to_monty((uint32_t)c));
    uint32_t h[8];
    host_hash(meta, h);
    host_compress(inner, h, h_commit);
    // what the evaluation proof (jagged.hip) needs later: JaggedProverData (prover.rs:L150-L156)
    (*out)->jagged = true;
    (*out)->max_log_row_count = max_log_row_count;
    (*out)->row_counts = rows;
    (*out)->column_counts = cols;
    (*out)->padding_column_count = added_cols;
    memcpy((*out)->jagged_commit, h_commit, 32);
    return SP1HIP_SUCCESS;
}

}  // extern "C"
