// Micro-benchmark: how long after the host publishes a value does a kernel start that was enqueued BEHIND a wait on that value
// (hipStreamWaitValue32: the command processor polls), compared with launching the kernel at that moment?
// The sumcheck rounds hand a challenge from the host to the next launch ~370 times per proof; the launch itself is on that path today.
//   hipcc --offload-arch=gfx950 -O2 bench/ubench/ubench_wait_value.hip -o /tmp/ubench_wait_value && /tmp/ubench_wait_value
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void mark(volatile uint32_t* done, uint32_t v, const uint32_t* in, uint32_t* out) {
    if (in) out[0] = in[0] + 1;                      // (reads what the host wrote before the flag: the challenge)
    __threadfence_system();
    *done = v;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *flag = nullptr, *done = nullptr, *chal = nullptr, *d_out = nullptr;
    if (hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) != hipSuccess) {
        (void)hipGetLastError();
        printf("no signal memory: the flag lives in mapped pinned host memory\n");
        CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
    }
    CK(hipHostMalloc((void**)&done, 64, hipHostMallocMapped));
    CK(hipHostMalloc((void**)&chal, 64, hipHostMallocMapped));
    CK(hipMalloc((void**)&d_out, 64));
    *flag = 0; *done = 0; *chal = 0;
    const int N = 300;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int mode = 0; mode < 3; mode++) {
        std::vector<double> lat;
        double api_wait = 0, api_launch = 0;
        for (int i = 1; i <= N; i++) {
            const uint32_t v = mode * 1000 + i;
            if (mode == 0) {                              // launch at the moment the value is known
                std::this_thread::sleep_for(std::chrono::microseconds(40));
                const auto t0 = now();
                *chal = v;
                hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, done, v, chal, d_out);
                while (*(volatile uint32_t*)done != v) {}
                lat.push_back(us(t0, now()));
            } else {                                       // pre-enqueued behind a wait; 1: wait >= v, 2: the same with 3 kernels queued behind it
                const auto ta = now();
                CK(hipStreamWaitValue32(s, flag, v, hipStreamWaitValueGte, 0xffffffffu));
                const auto tb = now();
                hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, done, v, chal, d_out);
                api_wait += us(ta, tb); api_launch += us(tb, now());
                std::this_thread::sleep_for(std::chrono::microseconds(40));
                const auto t0 = now();
                *chal = v;
                __atomic_store_n(flag, v, __ATOMIC_RELEASE);
                while (*(volatile uint32_t*)done != v) {}
                lat.push_back(us(t0, now()));
            }
        }
        CK(hipStreamSynchronize(s));
        std::sort(lat.begin(), lat.end());
        printf("mode %d (%s): median %.1f us, p10 %.1f, p90 %.1f\n", mode, mode == 0 ? "launch when known" : "pre-enqueued behind hipStreamWaitValue32",
               lat[N / 2], lat[N / 10], lat[N * 9 / 10]);
        if (mode == 1) printf("        host cost of the calls: hipStreamWaitValue32 %.1f us, the launch behind it %.1f us (means)\n", api_wait / N, api_launch / N);
        if (mode == 1) mode = 2;                           // (one waiting mode is enough)
    }
    return 0;
}
