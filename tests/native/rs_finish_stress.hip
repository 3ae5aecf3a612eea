// tests/native/rs_finish_stress.hip — stress of the library's sumcheck-round tail (`rs_finish`, sp1_amd/csrc/round_sync.hpp):
// the last workgroup to arrive reduces every workgroup's partial sums and publishes them to mapped host memory, with
// coherent (sc1) accesses + `s_waitcnt vmcnt(0)` instead of fences. Thousands of launches over grid sizes around the
// ticket's group boundaries, EVERY launch's sums checked on the host against field sums computed there; half of the
// launches are queued back to back (the counters must be left clean by the launch before). Workgroups also write 16 KiB
// of other output first, like a fold does. Prints "rs_finish stress: N launches, 0 wrong" and returns 0 when clean.
// Build (done by __graft_entry__.build()): hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isp1_amd/csrc -Iinclude ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "round_sync.hpp"

using namespace sp1hip;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

// a thread's contribution: (A(block) + B(thread, seq, s, k)) mod p, so the host can form the expected total in O(grid + 256)
__host__ __device__ inline uint32_t term_a(uint32_t block) { return (block * 2654435761u) % kb::P; }
__host__ __device__ inline uint32_t term_b(uint32_t thread, uint32_t seq, int s, int k) {
    return (thread * 40503u + seq * 2246822519u + (uint32_t)s * 7u + (uint32_t)k) % kb::P;
}
__host__ __device__ inline uint32_t term(uint32_t block, uint32_t thread, uint32_t seq, int s, int k) {
    return (term_a(block) + term_b(thread, seq, s, k)) % kb::P;
}

template <int NS>
__global__ __launch_bounds__(256) void tail_kernel(uint32_t* __restrict__ partials, RoundSync rs, uint32_t seq,
                                                   uint32_t* __restrict__ bulk, uint32_t bulk_words) {
    for (uint32_t i = threadIdx.x; i < bulk_words; i += 256) bulk[(size_t)blockIdx.x * bulk_words + i] = i ^ seq;
    kb::Ext acc[NS];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[s].c[k] = term(blockIdx.x, threadIdx.x, seq, s, k);
    rs_finish<NS>(acc, partials, blockIdx.x, gridDim.x, rs, seq);
}

// The other hand-over of the library that crosses workgroups on the way to the HOST (gkr.hip, a layer's last fold): every
// workgroup stores its rows straight into mapped host memory with system-scope stores, waits for their acknowledgement, takes
// a ticket; the last one publishes the sequence number. The host must then see EVERY row.
__global__ __launch_bounds__(256) void rows_kernel(uint32_t* host_rows, uint32_t n_rows, RoundSync rs, uint32_t seq) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r < n_rows)
        for (int k = 0; k < 16; k++) __hip_atomic_store(host_rows + (size_t)r * 16 + k, term_b(r, seq, k >> 2, k & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && rs_ticket_is_last(rs.counter, blockIdx.x, gridDim.x)) rs.host_slot[0] = seq;
}

template <int NS>
static void launch(uint32_t grid, uint32_t* d_partials, RoundSync rs, uint32_t seq, uint32_t* d_bulk, uint32_t bulk_words) {
    hipLaunchKernelGGL(tail_kernel<NS>, dim3(grid), dim3(256), 0, 0, d_partials, rs, seq, d_bulk, bulk_words);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 3000;
    const uint32_t max_grid = 5000, bulk_words = 4096;
    uint32_t *d_partials, *d_counter, *d_bulk, *h_slot, *d_slot;
    CHECK(hipMalloc(&d_partials, (size_t)max_grid * 4 * 10 * 4));
    CHECK(hipMalloc(&d_counter, RS_COUNTER_BYTES));
    CHECK(hipMemset(d_counter, 0, RS_COUNTER_BYTES));
    CHECK(hipMalloc(&d_bulk, (size_t)max_grid * bulk_words * 4));
    CHECK(hipHostMalloc(&h_slot, RS_SLOT_WORDS * 4, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void**)&d_slot, h_slot, 0));
    for (size_t i = 0; i < RS_SLOT_WORDS; i++) h_slot[i] = 0;
    const RoundSync rs{d_counter, (volatile uint32_t*)d_slot};
    const uint32_t grids[] = {1, 2, 3, 31, 32, 33, 63, 64, 65, 255, 256, 730, 1023, 1024, 1025, 4096, 5000};
    const int n_grids = sizeof(grids) / sizeof(grids[0]);
    const int nss[] = {1, 3, 4, 10};
    uint32_t seq = 0;
    long wrong = 0, checked = 0;
    for (int it = 0; it < rounds; it++) {
        const uint32_t grid = grids[it % n_grids];
        const int ns = nss[(it / n_grids) % 4];
        // every other iteration queues a burst of 3 launches back to back and checks the last one
        const int burst = (it & 1) ? 3 : 1;
        for (int b = 0; b < burst; b++) {
            ++seq;
            switch (ns) {
                case 1: launch<1>(grid, d_partials, rs, seq, d_bulk, bulk_words); break;
                case 3: launch<3>(grid, d_partials, rs, seq, d_bulk, bulk_words); break;
                case 4: launch<4>(grid, d_partials, rs, seq, d_bulk, bulk_words); break;
                default: launch<10>(grid, d_partials, rs, seq, d_bulk, bulk_words); break;
            }
        }
        CHECK(hipGetLastError());
        volatile uint32_t* slot = h_slot;
        uint64_t spins = 0;
        while (slot[0] != seq) {
            if ((++spins & 0xfffff) == 0 && hipStreamQuery(0) == hipSuccess && slot[0] != seq) {
                printf("rs_finish stress: launch %u (grid %u, NS %d) finished without publishing\n", seq, grid, ns);
                return 1;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        uint64_t sum_a = 0;
        for (uint32_t blk = 0; blk < grid; blk++) sum_a += term_a(blk);
        for (int s = 0; s < ns; s++)
            for (int k = 0; k < 4; k++) {
                uint64_t sum_b = 0;
                for (uint32_t t = 0; t < 256; t++) sum_b += term_b(t, seq, s, k);
                const uint64_t want = (256 * (sum_a % kb::P) + grid * (sum_b % kb::P)) % kb::P;
                checked++;
                if (slot[1 + 4 * s + k] != (uint32_t)(want % kb::P)) {
                    if (wrong < 5) printf("WRONG: launch %u grid %u NS %d sum %d.%d: %u != %u\n", seq, grid, ns, s, k, slot[1 + 4 * s + k], (uint32_t)(want % kb::P));
                    wrong++;
                }
            }
    }
    // ---- rows written by every workgroup straight to the host
    long rows_wrong = 0, rows_launches = 0;
    {
        const uint32_t max_rows = 1024;
        uint32_t *h_rows, *d_rows;
        CHECK(hipHostMalloc(&h_rows, (size_t)(max_rows * 16 + 16) * 4, hipHostMallocMapped));
        CHECK(hipHostGetDevicePointer((void**)&d_rows, h_rows, 0));
        const uint32_t row_counts[] = {1, 7, 255, 256, 257, 730, 1024};
        for (int it = 0; it < rounds; it++) {
            const uint32_t n_rows = row_counts[it % 7];
            ++seq;
            // the sequence number goes to word 0 of the block, the rows start 16 words in
            hipLaunchKernelGGL(rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, 0, d_rows + 16, n_rows,
                               RoundSync{d_counter, (volatile uint32_t*)d_rows}, seq);
            CHECK(hipGetLastError());
            volatile uint32_t* slot = h_rows;
            uint64_t spins = 0;
            while (slot[0] != seq)
                if ((++spins & 0xfffff) == 0 && hipStreamQuery(0) == hipSuccess && slot[0] != seq) { printf("rows: launch %u finished without publishing\n", seq); return 1; }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            for (uint32_t r = 0; r < n_rows; r++)
                for (int k = 0; k < 16; k++)
                    if (slot[16 + (size_t)r * 16 + k] != term_b(r, seq, k >> 2, k & 3)) { if (rows_wrong < 5) printf("WRONG ROW: launch %u row %u word %d\n", seq, r, k); rows_wrong++; }
            rows_launches++;
        }
    }
    CHECK(hipDeviceSynchronize());
    printf("direct rows: %ld launches, %ld wrong words\n", rows_launches, rows_wrong);
    wrong += rows_wrong;
    std::vector<uint32_t> counters(RS_COUNTER_BYTES / 4);
    CHECK(hipMemcpy(counters.data(), d_counter, RS_COUNTER_BYTES, hipMemcpyDeviceToHost));
    long dirty = 0;
    for (uint32_t c : counters) dirty += c != 0;
    printf("rs_finish stress: %u launches, %ld sums checked, %ld wrong, %ld counter words left non-zero\n", seq, checked, wrong, dirty);
    return wrong || dirty ? 1 : 0;
}
