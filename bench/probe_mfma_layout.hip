// prints the register layouts of v_mfma_f64_16x16x4_f64 and the gfx950 permlane swaps (used to write ubench_mfma.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, uint32_t* sw) {
    const int l = threadIdx.x, g = l >> 4, n = l & 15;
    d4 acc = {0, 0, 0, 0};
    for (int j = 0; j < 4; j++) {
        const double a = 100.0 * n + (4 * j + g);          // A_j[m = n][k = g] = 100 m + (4 j + k)
        const double b = (4 * j + g) == n ? 1.0 : 0.0;     // B_j[k = g][col = n] = identity
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) out[l * 4 + i] = acc[i];
    uint32_t a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    sw[l * 4 + 0] = r[0]; sw[l * 4 + 1] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    sw[l * 4 + 2] = q[0]; sw[l * 4 + 3] = q[1];
}
int main() {
    double* d; uint32_t* s;
    hipMalloc(&d, 256 * 8); hipMalloc(&s, 256 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, s);
    double h[256]; uint32_t hs[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof hs, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        printf("lane %2d: D =", l);
        for (int i = 0; i < 4; i++) printf(" (m=%d,n=%d)", (int)h[l * 4 + i] / 100, (int)h[l * 4 + i] % 100);
        printf("   swap32 -> a=%u b=%u   swap16 -> a=%u b=%u\n", hs[l * 4], hs[l * 4 + 1], hs[l * 4 + 2], hs[l * 4 + 3]);
    }
    return 0;
}
