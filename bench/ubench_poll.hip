// bench/ubench_poll.hip — what a sumcheck round's hand-over costs, by mechanism:
//   A  launch per round : host launches a tiny kernel that publishes a word to mapped pinned memory; host spins on it, then
//                         launches the next one (what the provers do today)
//   B  resident kernel  : ONE kernel is resident; each round the host writes a sequence number into mapped pinned memory,
//                         the kernel (polling it across PCIe) answers by writing the number back; host spins on the answer
// Printed: microseconds per round trip. B - (kernel-side work) is the floor of a "pre-enqueued kernel waits for its
// challenge" design; A is the launch + completion path it would replace.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 bench/ubench_poll.hip -o bench/ubench_poll
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <chrono>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void publish_kernel(volatile uint32_t* slot, uint32_t seq) { if (threadIdx.x == 0) slot[0] = seq; }

// answers `rounds` requests: waits until request[0] == r (bounded), writes r to answer[0]
__global__ void resident_kernel(const volatile uint32_t* request, volatile uint32_t* answer, uint32_t rounds, uint32_t max_spins) {
    if (threadIdx.x != 0) return;
    for (uint32_t r = 1; r <= rounds; r++) {
        uint32_t spins = 0;
        while (request[0] != r && ++spins < max_spins) __builtin_amdgcn_s_sleep(1);
        if (spins >= max_spins) { answer[1] = 0xdeadu; return; }
        answer[0] = r;
    }
}

int main() {
    uint32_t *h_req, *h_ans, *d_req, *d_ans;
    CHECK(hipHostMalloc(&h_req, 64, hipHostMallocMapped));
    CHECK(hipHostMalloc(&h_ans, 64, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void**)&d_req, h_req, 0));
    CHECK(hipHostGetDevicePointer((void**)&d_ans, h_ans, 0));
    volatile uint32_t* req = h_req;
    volatile uint32_t* ans = h_ans;
    req[0] = 0; ans[0] = 0; ans[1] = 0;
    const uint32_t rounds = 2000;
    // A: one launch per round
    for (int pass = 0; pass < 2; pass++) {
        ans[0] = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t r = 1; r <= rounds; r++) {
            hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, 0, d_ans, r);
            while (ans[0] != r) {}
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (pass) printf("A  launch per round         : %6.2f us per round trip\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds);
    }
    CHECK(hipDeviceSynchronize());
    // B: resident kernel polling host memory
    for (int pass = 0; pass < 2; pass++) {
        req[0] = 0; ans[0] = 0;
        hipLaunchKernelGGL(resident_kernel, dim3(1), dim3(64), 0, 0, d_req, d_ans, rounds, 50000000u);
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        for (uint32_t r = 1; r <= rounds && ok; r++) {
            req[0] = r;
            const auto w0 = std::chrono::steady_clock::now();
            while (ans[0] != r) {
                if (ans[1] == 0xdeadu || std::chrono::steady_clock::now() - w0 > std::chrono::seconds(2)) { ok = false; break; }
            }
        }
        const auto t1 = std::chrono::steady_clock::now();
        CHECK(hipDeviceSynchronize());
        if (pass) printf("B  resident kernel, polling : %6.2f us per round trip %s\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds, ok ? "" : "(TIMED OUT)");
    }
    return 0;
}
