// bench/ubench_cycles.hip — issue cost (shader cycles per wave-instruction) of the integer VALU
// instructions the KoalaBear/Poseidon2 code is built from, measured with s_memtime on gfx950.
// One workgroup of 256 threads per CU (1 wave per SIMD) and, second column, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 bench/ubench_cycles.hip -o bench/ubench_cycles
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 512;
constexpr int ILP = 8;

#define OPS(X) X(ADD) X(SUB) X(MIN) X(LSHL_ADD) X(ADD3) X(MUL_LO) X(MUL_HI) X(MAD_U64) X(MAD_I64) X(MUL_U24) \
    X(LSHL_ADD_U64) X(LSHLREV_B64) X(ADDC64) X(PAIR_MUL_ADD) X(TRIPLE_MUL_ADD_ADD) X(MM_A) X(MM_B) X(MM_SIGNED) X(ADDMOD) X(MOV)
enum Op {
#define X(n) n,
    OPS(X)
#undef X
    N_OPS
};
static const char* NAMES[] = {
#define X(n) #n,
    OPS(X)
#undef X
};
static const int INSTRS[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 6, 5, 4, 3, 1};

constexpr uint32_t P = 0x7f000001u, MU = 0x81000001u, NMU = 0x7effffffu;

template <int OP>
__device__ __forceinline__ void body(uint32_t (&x)[ILP], uint64_t (&w)[ILP], uint32_t y) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
        if (OP == ADD) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == SUB) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == MIN) asm volatile("v_min_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == ADD3) asm volatile("v_add3_u32 %0, %1, %2, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(x[i]), "v"(y) : "vcc");
        if (OP == MAD_I64) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(x[i]), "v"(y) : "vcc");
        if (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
        if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(w[i]) : "v"(w[(i + 1) % ILP]));
        if (OP == LSHLREV_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w[i]));
        if (OP == ADDC64) {
            uint32_t lo = (uint32_t)w[i], hi = (uint32_t)(w[i] >> 32);
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(x[i]), "v"(y) : "vcc");
            w[i] = ((uint64_t)hi << 32) | lo;
        }
        if (OP == PAIR_MUL_ADD) {
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            uint32_t lo = (uint32_t)w[i];
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(y));
            w[i] = lo;
        }
        if (OP == TRIPLE_MUL_ADD_ADD) {
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y));
            uint32_t lo = (uint32_t)w[i], hi = (uint32_t)(w[i] >> 32);
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(y));
            asm volatile("v_min_u32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(y));
            w[i] = ((uint64_t)hi << 32) | lo;
        }
        if (OP == MM_A) {
            uint64_t ab = (uint64_t)x[i] * x[i];
            uint32_t t = (uint32_t)ab * MU, u = __umulhi(t, P), r = (uint32_t)(ab >> 32) - u, r2 = r + P;
            x[i] = r < r2 ? r : r2;
        }
        if (OP == MM_B) {
            uint64_t ab = (uint64_t)x[i] * x[i];
            uint32_t t = (uint32_t)ab * NMU;
            uint64_t s = (uint64_t)t * P + ab;
            uint32_t r = (uint32_t)(s >> 32), r2 = r - P;
            x[i] = r < r2 ? r : r2;
        }
        if (OP == MM_SIGNED) {  // lazy signed form: mad_i64, mul_lo, mul_hi, sub (no correction)
            int64_t ab = (int64_t)(int32_t)x[i] * (int32_t)x[i];
            uint32_t t = (uint32_t)ab * MU, u = __umulhi(t, P);
            x[i] = (uint32_t)(ab >> 32) - u;
        }
        if (OP == ADDMOD) { uint32_t s = x[i] + y, s2 = s - P; x[i] = s < s2 ? s : s2; }
        if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y));
    }
}

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* cycles, uint32_t* sink, uint32_t seed) {
    uint32_t x[ILP], y = seed | 1u;
    uint64_t w[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { x[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) % P; w[i] = x[i]; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) body<OP>(x, w, y);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc ^= x[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
int run(uint64_t* d_cyc, uint32_t* d_sink, int cus) {
    double res[2];
    for (int mode = 0; mode < 2; mode++) {
        const int blocks = cus * (mode ? 8 : 1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 7u);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 9u);
        CHECK(hipDeviceSynchronize());
        static uint64_t h[8 * 256 * 4];
        CHECK(hipMemcpy(h, d_cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int i = 0; i < blocks * 4; i++) sum += (double)h[i];
        double per_wave = sum / (blocks * 4) / ((double)ITERS * ILP);
        // with W waves per SIMD sharing the issue port, cost per instruction = elapsed / (W * instr)
        res[mode] = per_wave / (mode ? 8 : 1);
    }
    printf("%-20s instr=%d  cycles/op: 1 wave/SIMD %7.2f   8 waves/SIMD %7.2f   (per instr %5.2f / %5.2f)\n", NAMES[OP],
           INSTRS[OP], res[0], res[1], res[0] / INSTRS[OP], res[1] / INSTRS[OP]);
    return 0;
}

template <int OP>
struct Runner {
    static int go(uint64_t* c, uint32_t* s, int cus) {
        if (run<OP>(c, s, cus)) return 1;
        return Runner<OP + 1>::go(c, s, cus);
    }
};
template <>
struct Runner<N_OPS> {
    static int go(uint64_t*, uint32_t*, int) { return 0; }
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%d CUs; s_memtime ticks per wave-instruction (ILP %d)\n", cus, ILP);
    uint64_t* d_cyc;
    uint32_t* d_sink;
    CHECK(hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * 8));
    CHECK(hipMalloc(&d_sink, (size_t)cus * 8 * 256 * 4));
    return Runner<0>::go(d_cyc, d_sink, cus);
}
