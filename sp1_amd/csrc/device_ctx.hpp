// sp1_amd/csrc/device_ctx.hpp — per-GPU immutable tables shared by all kernels of libsp1hip.so:
// Poseidon2 round constants (Montgomery form) and the NTT twiddle tables. Built lazily, once per
// device, and kept for the life of the process (one process per GPU in the multi-GPU layout).
#pragma once
#include "common.hpp"
#include "poseidon2.hpp"

namespace sp1hip {

// Twiddle tables for the two-level power lookup used by the NTT and fold kernels:
//   w_{2^24}^e  =  hi[e >> 12] * lo[e & 4095],  hi[j] = w^(4096 j), lo[j] = w^j      (4096 + 4096 words)
// Any w_{2^k}^e (k <= 24) is w_{2^24}^(e << (24 - k)).
constexpr int TW_LO_BITS = 12;
constexpr int TW_LO = 1 << TW_LO_BITS;
constexpr int TW_HI = 1 << (kb::TWO_ADICITY - TW_LO_BITS);

struct DeviceCtx {
    int device = -1;
    p2::RoundConstants* d_rc = nullptr;
    uint32_t* d_tw_lo = nullptr;  // [TW_LO]
    uint32_t* d_tw_hi = nullptr;  // [TW_HI]
    int num_cus = 256;
};

// Returns the context of the CURRENT device (hipGetDevice), creating it on first use.
int get_device_ctx(const DeviceCtx** out);

}  // namespace sp1hip
