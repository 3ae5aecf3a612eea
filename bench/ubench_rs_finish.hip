// bench/ubench_rs_finish.hip — what the "last workgroup reduces and publishes" tail of a sumcheck round costs, and how
// much of it is the agent-scope release / acquire fence pair (gfx950: buffer_wbl2 sc1 / buffer_inv sc1 — every
// workgroup writes back its XCD's L2) against coherent (sc1) stores and loads of the few words that actually cross
// workgroups.
//   A  fence protocol   : plain stores of the partials, s_waitcnt, barrier, release fence, ticket; last: acquire, plain loads
//   B  coherent accesses: sc1 stores of the partials, s_waitcnt, barrier, ticket; last: sc1 loads; no fence
//   C  B with a two-level ticket (16 group counters on separate lines, then one): same-address atomics serialise
// Each variant with the workgroups writing `dirty` KiB of other output first (the folded tables of a round).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isp1_amd/csrc bench/ubench_rs_finish.hip -o bench/ubench_rs_finish
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NW = 12;

constexpr uint32_t GROUPS = 16, GROUP_STRIDE = 64;      // counters of the two-level ticket: word g * GROUP_STRIDE, level 2 at GROUPS * GROUP_STRIDE

template <bool COHERENT, bool TWO_LEVEL = false>
__global__ __launch_bounds__(256) void tail_kernel(uint32_t* __restrict__ partials, uint32_t* counter, volatile uint32_t* host_slot,
                                                   uint32_t seq, uint32_t* __restrict__ bulk, uint32_t bulk_words) {
    __shared__ uint32_t last_flag;
    __shared__ uint32_t sm[4][NW];
    for (uint32_t i = threadIdx.x; i < bulk_words; i += 256) bulk[(size_t)blockIdx.x * bulk_words + i] = i + seq;
    uint32_t v = threadIdx.x + blockIdx.x + seq;
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < NW; k++) sm[threadIdx.x >> 6][k] = v + k;
    __syncthreads();
    if (threadIdx.x < NW) {
        const uint32_t a = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
        if (COHERENT) __hip_atomic_store(&partials[(size_t)blockIdx.x * NW + threadIdx.x], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partials[(size_t)blockIdx.x * NW + threadIdx.x] = a;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!COHERENT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (TWO_LEVEL) {
            const uint32_t g = blockIdx.x % GROUPS, members = (gridDim.x - g + GROUPS - 1) / GROUPS;
            uint32_t last = 0;
            if (__hip_atomic_fetch_add(counter + g * GROUP_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(counter + g * GROUP_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t groups = gridDim.x < GROUPS ? gridDim.x : GROUPS;
                last = __hip_atomic_fetch_add(counter + GROUPS * GROUP_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
                if (last) __hip_atomic_store(counter + GROUPS * GROUP_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            last_flag = last;
        } else {
        const uint32_t ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = ticket == gridDim.x - 1;
        }
        if (!COHERENT && last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!last_flag) return;
    uint32_t tot[NW];
    for (int k = 0; k < NW; k++) tot[k] = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += 256)
        for (int k = 0; k < NW; k++)
            tot[k] += COHERENT ? __hip_atomic_load(&partials[(size_t)i * NW + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : partials[(size_t)i * NW + k];
    __syncthreads();
    for (int k = 0; k < NW; k++) {
        uint32_t w = tot[k];
        for (int off = 32; off >= 1; off >>= 1) w += __shfl_down(w, off, 64);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < NW) host_slot[1 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    if (threadIdx.x == 0 && !TWO_LEVEL) {
        if (COHERENT) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *counter = 0;
    }
    if (COHERENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) host_slot[0] = seq;
}

__global__ void empty_kernel() {}

int main() {
    uint32_t *d_partials, *d_counter, *d_bulk, *h_slot, *d_slot;
    const int max_blocks = 8192;
    const uint32_t max_bulk = 16384;      // words per block
    CHECK(hipMalloc(&d_partials, (size_t)max_blocks * NW * 4));
    CHECK(hipMalloc(&d_counter, 8192));
    CHECK(hipMemset(d_counter, 0, 8192));
    CHECK(hipMalloc(&d_bulk, (size_t)max_blocks * max_bulk * 4));
    CHECK(hipHostMalloc(&h_slot, 128, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void**)&d_slot, h_slot, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 200;
    uint32_t seq = 0;
    {
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(empty_kernel, dim3(730), dim3(256), 0, 0);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel, 730 workgroups: %.2f us per launch (back to back)\n", ms * 1e3 / reps);
    }
    const int grids[] = {64, 730, 1024, 4096};
    const uint32_t bulks[] = {0, 1024, 16384};
    for (int grid : grids)
        for (uint32_t bulk : bulks)
            for (int variant = 0; variant < 3; variant++) {
                float ms = 0;
                for (int pass = 0; pass < 2; pass++) {
                    CHECK(hipEventRecord(e0));
                    for (int i = 0; i < reps; i++) {
                        ++seq;
                        if (variant == 0) hipLaunchKernelGGL(tail_kernel<false>, dim3(grid), dim3(256), 0, 0, d_partials, d_counter, d_slot, seq, d_bulk, bulk);
                        else if (variant == 1) hipLaunchKernelGGL(tail_kernel<true>, dim3(grid), dim3(256), 0, 0, d_partials, d_counter, d_slot, seq, d_bulk, bulk);
                        else hipLaunchKernelGGL((tail_kernel<true, true>), dim3(grid), dim3(256), 0, 0, d_partials, d_counter, d_slot, seq, d_bulk, bulk);
                    }
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                // expected total of word 0: sum over blocks of (sum over 256 threads of (t + b + seq))
                uint64_t want = 0;
                for (int b = 0; b < grid; b++) want += (uint64_t)256 * (b + seq) + 255 * 128;
                const bool ok = h_slot[0] == seq && h_slot[1] == (uint32_t)want;
                printf("grid %5d  bulk %3u KiB/wg  %s : %7.2f us per launch   %s\n", grid, bulk * 4 / 1024,
                       variant == 0 ? "A fences        " : variant == 1 ? "B coherent sc1  " : "C B + 2-lvl tick", ms * 1e3 / reps, ok ? "ok" : "WRONG RESULT");
            }
    return 0;
}
