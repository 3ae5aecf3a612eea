// bench/ubench_int.hip — gfx950 integer-VALU micro-benchmarks that size the KoalaBear arithmetic:
// issue rate of the 32-bit multiply family vs plain adds, and whole Montgomery-multiply variants.
// Build: hipcc --offload-arch=gfx950 -O3 bench/ubench_int.hip -o bench/ubench_int
// Prints giga-ops/s over the whole chip; divide by (CUs * 4 SIMDs * clock) for lanes/clk/SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

enum Op { ADD, MIN, LSHL_ADD, MUL_LO, MUL_HI, MAD64, MUL24, MAD24, MM_A, MM_B, MM_C, ADDMOD };

constexpr uint32_t P = 0x7f000001u, MU = 0x81000001u, NMU = 0x7effffffu;

__device__ __forceinline__ uint32_t mm_a(uint32_t a, uint32_t b) {  // mad64 + mul_lo + mul_hi + sub/add/min
    uint64_t ab = (uint64_t)a * b;
    uint32_t t = (uint32_t)ab * MU;
    uint32_t u = __umulhi(t, P);
    uint32_t r = (uint32_t)(ab >> 32) - u;
    uint32_t r2 = r + P;
    return r < r2 ? r : r2;
}
__device__ __forceinline__ uint32_t mm_b(uint32_t a, uint32_t b) {  // two mad64, quotient digit by multiply
    uint64_t ab = (uint64_t)a * b;
    uint32_t t = (uint32_t)ab * NMU;              // -p^-1
    uint64_t s = (uint64_t)t * P + ab;            // low word cancels
    uint32_t r = (uint32_t)(s >> 32);
    uint32_t r2 = r - P;
    return r < r2 ? r : r2;
}
__device__ __forceinline__ uint32_t mm_c(uint32_t a, uint32_t b) {  // quotient digit by shifts (forced)
    uint64_t ab = (uint64_t)a * b;
    uint32_t lo = (uint32_t)ab, t1, t;
    asm volatile("v_lshl_add_u32 %0, %1, 24, %1" : "=v"(t1) : "v"(lo));
    asm volatile("v_lshl_add_u32 %0, %1, 31, %2" : "=v"(t) : "v"(lo), "v"(t1));
    uint32_t u = __umulhi(t, P);
    uint32_t r = (uint32_t)(ab >> 32) - u;
    uint32_t r2 = r + P;
    return r < r2 ? r : r2;
}

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP], y = seed | 1u;
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) % P;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == ADD) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            if (OP == MIN) asm volatile("v_min_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            if (OP == MAD64) {
                uint64_t r;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(x[i]), "v"(y), "v"((uint64_t)x[i]) : "vcc");
                x[i] = (uint32_t)(r >> 32) ^ (uint32_t)r;
            }
            if (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            if (OP == MM_A) x[i] = mm_a(x[i], x[i]);
            if (OP == MM_B) x[i] = mm_b(x[i], x[i]);
            if (OP == MM_C) x[i] = mm_c(x[i], x[i]);
            if (OP == ADDMOD) { uint32_t s = x[i] + y; uint32_t s2 = s - P; x[i] = s < s2 ? s : s2; }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
int run(const char* name, uint32_t* d_out, int blocks) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    double ops = 5.0 * blocks * 256.0 * ITERS * ILP;
    printf("%-10s %10.1f Gop/s   (%.3f ms/launch)\n", name, ops / (ms * 1e-3) / 1e9, ms / 5);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = prop.multiProcessorCount * 8;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    run<ADD>("add", d_out, blocks);
    run<MIN>("min", d_out, blocks);
    run<LSHL_ADD>("lshl_add", d_out, blocks);
    run<MUL_LO>("mul_lo", d_out, blocks);
    run<MUL_HI>("mul_hi", d_out, blocks);
    run<MAD64>("mad_u64", d_out, blocks);
    run<MUL24>("mul_u24", d_out, blocks);
    run<MAD24>("mad_u24", d_out, blocks);
    run<ADDMOD>("addmod", d_out, blocks);
    run<MM_A>("montmul_A", d_out, blocks);
    run<MM_B>("montmul_B", d_out, blocks);
    run<MM_C>("montmul_C", d_out, blocks);
    return 0;
}
