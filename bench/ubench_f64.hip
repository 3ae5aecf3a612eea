// bench/ubench_f64.hip — gfx950 issue rates of the fp64 / conversion / signed-mad instructions that an
// "exact fp64 linear layer" formulation of Poseidon2's external rounds would lean on, next to v_add_u32.
// Build: hipcc --offload-arch=gfx950 -O3 bench/ubench_f64.hip -o bench/ubench_f64
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;
enum Op { ADD_U32, ADD_F64, MUL_F64, FMA_F64, RNDNE_F64, CVT_RT, MAD_I64, ADD3, ADD_CO64, REDUCE_F64, ADDMOD };

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t xi[ILP], yi = seed | 1u;
    double xd[ILP], yd = 1.000000001 + seed * 1e-12, zd = 3.0e-9;
#pragma unroll
    for (int i = 0; i < ILP; i++) { xi[i] = threadIdx.x * 2654435761u + i * 40503u + seed; xd[i] = (double)xi[i]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == ADD_U32) asm volatile("v_add_u32 %0, %1, %2" : "=v"(xi[i]) : "v"(xi[i]), "v"(yi));
            if (OP == ADD_F64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(xd[i]) : "v"(xd[i]), "v"(yd));
            if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(xd[i]) : "v"(xd[i]), "v"(yd));
            if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(xd[i]) : "v"(xd[i]), "v"(yd), "v"(zd));
            if (OP == RNDNE_F64) asm volatile("v_rndne_f64 %0, %1" : "=v"(xd[i]) : "v"(xd[i]));
            if (OP == CVT_RT) {                      // two instructions per iteration: i32 -> f64 -> i32
                double t;
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(t) : "v"(xi[i]));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(xi[i]) : "v"(t));
            }
            if (OP == MAD_I64) {                     // two instructions: mad_i64_i32 + xor (like ubench_int's mad_u64)
                uint64_t r;
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(xi[i]), "v"(yi), "v"((uint64_t)xi[i]) : "vcc");
                xi[i] = (uint32_t)(r >> 32) ^ (uint32_t)r;
            }
            if (OP == ADD3) asm volatile("v_add3_u32 %0, %1, %2, %2" : "=v"(xi[i]) : "v"(xi[i]), "v"(yi));
            if (OP == ADD_CO64) {                    // a 64-bit integer add = add_co + addc_co
                uint64_t a = ((uint64_t)xi[i] << 32) | xi[(i + 1) % ILP], b = ((uint64_t)yi << 32) | yi;
                a += b;
                xi[i] = (uint32_t)(a >> 32) + (uint32_t)a;
            }
            if (OP == REDUCE_F64) {                  // x -> x - rndne(x / p) * p, then to int and back
                double q, r;
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(q) : "v"(xd[i]), "v"(zd));
                asm volatile("v_rndne_f64 %0, %1" : "=v"(q) : "v"(q));
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(q), "v"(yd), "v"(xd[i]));
                xd[i] = r;
            }
            if (OP == ADDMOD) { uint32_t s = xi[i] + yi; uint32_t s2 = s - 0x7f000001u; xi[i] = s < s2 ? s : s2; }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc ^= xi[i] ^ (uint32_t)(long long)xd[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
int run(const char* name, uint32_t* d_out, int blocks, double insts) {
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
    double iters = 5.0 * blocks * 256.0 * ITERS * ILP;
    printf("%-12s %10.1f G iter/s  %10.1f G inst/s  (%.3f ms/launch, %g inst/iter)\n", name, iters / (ms * 1e-3) / 1e9,
           iters * insts / (ms * 1e-3) / 1e9, ms / 5, insts);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = prop.multiProcessorCount * 8;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    run<ADD_U32>("add_u32", d_out, blocks, 1);
    run<ADD3>("add3_u32", d_out, blocks, 1);
    run<ADDMOD>("addmod", d_out, blocks, 3);
    run<ADD_CO64>("add_u64", d_out, blocks, 3);
    run<ADD_F64>("add_f64", d_out, blocks, 1);
    run<MUL_F64>("mul_f64", d_out, blocks, 1);
    run<FMA_F64>("fma_f64", d_out, blocks, 1);
    run<RNDNE_F64>("rndne_f64", d_out, blocks, 1);
    run<CVT_RT>("cvt_i32<->f64", d_out, blocks, 2);
    run<MAD_I64>("mad_i64+xor", d_out, blocks, 2);
    run<REDUCE_F64>("reduce_f64", d_out, blocks, 3);
    return 0;
}
