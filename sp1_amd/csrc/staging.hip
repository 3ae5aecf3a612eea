// sp1_amd/csrc/staging.hip — host traces -> device tables (the staging half).
//
// The reference's CPU chips write their traces row-major (`[height][width]`, `generate_trace_into`) into one
// pinned host buffer; `device_main_tracegen` then copies every trace to the device and transposes it
// (/root/reference/sp1-gpu/crates/jagged_tracegen/src/lib.rs:L719-L835: "copy host trace to device",
// `DeviceTensor::from_raw(tensor).transpose()`), one blocking copy + one transpose kernel per chip.
//
// MI355X shape: the copy is the slow part (PCIe, ~50 GB/s against 8 TB/s of HBM), so it must never wait for
// anything. All tables are cut into row chunks that fit one of two device staging buffers; the SDMA copy of
// chunk i + 1 runs on a side stream while the caller's stream transposes chunk i straight into its final
// column-major place (64 x 64 LDS tiles, 256 B runs on both sides). The caller's stream never blocks the
// host: stage shard k + 1 on one stream while another stream proves shard k.
#include "device_ctx.hpp"
#include "tensor_table.hpp"

namespace sp1hip {

// in: row-major [n_rows][cols] chunk; out: column-major table with `ld_out` rows per column, chunk starts at row `row0`.
__global__ __launch_bounds__(256) void stage_transpose_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                              uint32_t n_rows, uint32_t cols, uint64_t ld_out, uint64_t row0) {
    __shared__ uint32_t tile[64][65];
    const uint32_t r0 = blockIdx.x * 64u, c0 = blockIdx.y * 64u;   // row tiles on x: up to 2^17 of them per chunk
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
#pragma unroll 4
    for (uint32_t k = w; k < 64; k += 4) {
        const uint32_t r = r0 + k, c = c0 + lane;
        if (r < n_rows && c < cols) tile[k][lane] = in[(uint64_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t k = w; k < 64; k += 4) {
        const uint32_t c = c0 + k, r = r0 + lane;
        if (r < n_rows && c < cols) out[(uint64_t)c * ld_out + row0 + r] = tile[lane][k];
    }
}

constexpr size_t STAGE_BYTES = (size_t)32 << 20;   // per staging buffer; two in flight

}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_stage_tables(const sp1hip_host_table_t* tables, int n_tables, uint32_t* const* d_out, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE((tables && d_out) || n_tables == 0, "null argument");
    SP1HIP_REQUIRE(n_tables >= 0, "negative table count");
    uint64_t total = 0;
    for (int i = 0; i < n_tables; i++) {
        const uint64_t n = tables[i].rows * (uint64_t)tables[i].cols;
        SP1HIP_REQUIRE(n == 0 || (tables[i].h_data && d_out[i]), "null table data");
        SP1HIP_REQUIRE(tables[i].rows < ((uint64_t)1 << 32), "table taller than 2^32 rows");
        SP1HIP_REQUIRE((uint64_t)tables[i].cols * 4 * 64 <= STAGE_BYTES, "table too wide for the staging buffer");
        total += n;
    }
    if (total == 0) return SP1HIP_SUCCESS;
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    hipStream_t s = S(stream), aux;
    hipEvent_t* ev;
    SP1HIP_TRY(aux_stream_for(s, 5, &aux, &ev));      // [0,1] copied, [2,3] consumed, [4] fork
    AsyncScratch stage[2];
    SP1HIP_TRY(stage[0].alloc(STAGE_BYTES, s));
    SP1HIP_TRY(stage[1].alloc(STAGE_BYTES, s));
    struct Join {                                     // never hand the staging buffers back while a copy is in flight
        hipStream_t aux;
        hipStream_t s;
        hipEvent_t* ev;
        uint64_t chunks = 0;
        ~Join() {
            // the caller's stream already waits for every copy it consumed; an early error return may leave one
            // un-consumed copy on the side stream
            (void)hipEventRecord(ev[4], aux);
            (void)hipStreamWaitEvent(s, ev[4], 0);
        }
    } join{aux, s, ev};
    SP1HIP_HIP(hipEventRecord(ev[4], s));             // recycled staging blocks / output buffers are ordered on `s`
    SP1HIP_HIP(hipStreamWaitEvent(aux, ev[4], 0));
    ScopedTimer timer("stage_tables", s);
    for (int i = 0; i < n_tables; i++) {
        const uint64_t rows = tables[i].rows;
        const uint32_t cols = tables[i].cols;
        if (rows == 0 || cols == 0) continue;
        uint64_t per = (STAGE_BYTES / ((uint64_t)cols * 4)) & ~(uint64_t)63;      // whole 64-row tiles per chunk
        for (uint64_t r0 = 0; r0 < rows; r0 += per) {
            const uint64_t nr = rows - r0 < per ? rows - r0 : per;
            const int b = (int)(join.chunks & 1);
            if (join.chunks >= 2) SP1HIP_HIP(hipStreamWaitEvent(aux, ev[2 + b], 0));
            SP1HIP_HIP(hipMemcpyAsync(stage[b].p, tables[i].h_data + r0 * cols, nr * cols * 4, hipMemcpyHostToDevice, aux));
            SP1HIP_HIP(hipEventRecord(ev[b], aux));
            SP1HIP_HIP(hipStreamWaitEvent(s, ev[b], 0));
            const dim3 grid((unsigned)((nr + 63) / 64), (cols + 63) / 64);
            hipLaunchKernelGGL(stage_transpose_kernel, grid, dim3(256), 0, s, (const uint32_t*)stage[b].p, d_out[i],
                               (uint32_t)nr, cols, rows, r0);
            SP1HIP_LAUNCH_CHECK();
            SP1HIP_HIP(hipEventRecord(ev[2 + b], s));
            join.chunks++;
        }
    }
    return SP1HIP_SUCCESS;
}

int sp1hip_host_register(void* h_ptr, size_t bytes) {
    SP1HIP_REQUIRE(h_ptr || bytes == 0, "null pointer");
    if (bytes == 0) return SP1HIP_SUCCESS;
    SP1HIP_HIP(hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return SP1HIP_SUCCESS;
}
int sp1hip_host_unregister(void* h_ptr) {
    SP1HIP_REQUIRE(h_ptr, "null pointer");
    SP1HIP_HIP(hipHostUnregister(h_ptr));
    return SP1HIP_SUCCESS;
}

}  // extern "C"
