// host-side cost of a kernel launch on this box (empty kernel, 1000 launches back to back), with and without a large kernarg
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
struct Big { unsigned w[60]; };
__global__ void empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void empty_big(int* p, Big b) { if (p && threadIdx.x == 9999) *p = b.w[3]; }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    int* d; hipMalloc(&d, 4);
    Big b{};
    for (int big = 0; big < 2; big++) {
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(empty, dim3(730), dim3(256), 0, s, d);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 1000; i++) { if (big) hipLaunchKernelGGL(empty_big, dim3(730), dim3(256), 0, s, d, b); else hipLaunchKernelGGL(empty, dim3(730), dim3(256), 0, s, d); }
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        printf("%s kernarg: enqueue %.2f us per launch, end-to-end %.2f us per launch\n", big ? "240-byte" : "small",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / 1000, std::chrono::duration<double, std::micro>(t2 - t0).count() / 1000);
    }
    return 0;
}
