// bench/ubench_mfma.hip — Poseidon2-KoalaBear EXTERNAL rounds with the 16x16 linear layer on the matrix pipe
// (v_mfma_f64_16x16x4_f64) against the production form (64 v_add_f64 / v_fma_f64 per lane), VERDICT r1 #7(c).
//
// The external layer is d' = M d with M the circulant-of-M4 16x16 integer matrix (entries 1..6), and the state is
// held as exact doubles (poseidon2.hpp), so an f64 MFMA computes it EXACTLY (|entries| < 2^44, sums of 16 products
// of a <= 6 coefficient: < 2^51).
//   A (VALU):  one permutation per lane, d[16] per lane; a round = 16 S-boxes (11 VALU each) + the add tree.
//   B (MFMA):  one permutation per FOUR lanes: lane l holds words (l / 16) + 4 i, i = 0..3, of state l % 16 — the
//              C/D layout of a 16x16 f64 tile (probed with bench/probe_mfma_layout.hip), which is ALSO the B-operand
//              layout of the four k-slices: the result of one layer is in place for the S-box and for the next
//              layer, no shuffle at all. A layer = 4 MFMAs accumulating D = M X; S-box: 4 per lane. This is the
//              most favourable layout the matrix pipe can be given (the 20 internal rounds in between would still
//              need the one-permutation-per-lane layout, i.e. two transposes per permutation that are NOT charged here).
// Per 16 states and round: A = 60 VALU; B = 44 VALU and 4 MFMA. Both variants run ROUNDS rounds over the same
// states and must produce identical integers (checked on the host). Printed: ns per state-round and the ratio; run
// under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA` for the pipe occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isp1_amd/csrc bench/ubench_mfma.hip -o bench/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "poseidon2.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ROUNDS = 256;          // external rounds per launch (the 8 round-constant rows are cycled)
typedef double d4 __attribute__((ext_vector_type(4)));

// ---- A: production form ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ext_valu(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                const p2::RoundConstants* __restrict__ rc) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = (double)in[t * 16 + i];
#pragma unroll 1
    for (int r = 0; r < ROUNDS; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = p2::sbox_f64(d[i], rc->ext_magic[r & 7][i]);
        p2::external_linear_f64(d);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[t * 16 + i] = p2::canonical_f64(d[i]);
}

// ---- B: linear layer on the matrix pipe ------------------------------------------------------------------------
// entry (row, col) of the external matrix: M4 circulant blocks, + the block pattern of the column sums
__host__ __device__ inline double mds(int row, int col) {
    const int m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    const int v = m4[row & 3][col & 3];
    return (double)((row >> 2) == (col >> 2) ? 2 * v : v);
}

__global__ __launch_bounds__(256) void ext_mfma(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                const p2::RoundConstants* __restrict__ rc) {
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t state = wave * 16 + n;                       // 16 states per wave
    double x[4];                                              // words g + 4 i of state n
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = (double)in[state * 16 + g + 4 * i];
    // A operand of k-slice j: lane l holds M[row = l % 16][col = 4 j + l / 16]
    double a[4];
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = mds(n, 4 * j + g);
#pragma unroll 1
    for (int r = 0; r < ROUNDS; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = p2::sbox_f64(x[i], rc->ext_magic[r & 7][g + 4 * i]);
        // k-slice j wants B[k = l / 16][col] = word 4 j + g of the state = x[j]: already in place
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], x[j], acc, 0, 0, 0);
        // D layout (probed on gfx950, bench/probe_mfma_layout.hip): lane l, register i = row (l / 16) + 4 i of column l % 16
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = acc[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[state * 16 + g + 4 * i] = p2::canonical_f64(x[i]);
}

int main(int argc, char** argv) {
    const size_t n_states = (argc > 1 ? (size_t)atol(argv[1]) : (size_t)1 << 22);   // multiple of 1024
    const p2::RoundConstants h_rc = p2::make_round_constants();
    p2::RoundConstants* d_rc;
    CHECK(hipMalloc(&d_rc, sizeof h_rc));
    CHECK(hipMemcpy(d_rc, &h_rc, sizeof h_rc, hipMemcpyHostToDevice));
    std::vector<uint32_t> h_in(n_states * 16);
    uint64_t s = 42;
    for (auto& w : h_in) { s = s * 6364136223846793005ull + 1442695040888963407ull; w = (uint32_t)((s >> 33) % kb::P); }
    uint32_t *d_in, *d_a, *d_b;
    CHECK(hipMalloc(&d_in, h_in.size() * 4));
    CHECK(hipMalloc(&d_a, h_in.size() * 4));
    CHECK(hipMalloc(&d_b, h_in.size() * 4));
    CHECK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms_a = 0, ms_b = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(ext_valu, dim3(n_states / 256), dim3(256), 0, 0, d_in, d_a, d_rc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_a, e0, e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(ext_mfma, dim3(n_states * 4 / 256), dim3(256), 0, 0, d_in, d_b, d_rc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_b, e0, e1));
    }
    std::vector<uint32_t> ha(h_in.size()), hb(h_in.size());
    CHECK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ha.size(); i++) bad += ha[i] != hb[i];
    const double sr = (double)n_states * ROUNDS;
    printf("states %zu, %d external rounds each; results %s (%zu words differ)\n", n_states, ROUNDS, bad ? "DIFFER" : "identical", bad);
    printf("A  VALU linear layer (1 state / lane)      : %8.3f ms  %7.4f ns per state-round  (%d VALU per state-round)\n", ms_a, ms_a * 1e6 / sr, 240);
    printf("B  MFMA f64 16x16x4 layer (1 state / 4 lanes): %8.3f ms  %7.4f ns per state-round  (44 VALU + 4 MFMA per 16 states)\n", ms_b, ms_b * 1e6 / sr);
    printf("B / A = %.3f\n", ms_b / ms_a);
    return bad ? 2 : 0;
}
