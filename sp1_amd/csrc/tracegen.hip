// sp1_amd/csrc/tracegen.hip — trace generation ON THE DEVICE for the recursion machine.
//
// The reference's CUDA backend generates the main traces of the recursion chips on the GPU from the execution record's
// event arrays (`CudaTracegenAir::generate_trace_device`, /root/reference/sp1-gpu/crates/tracegen/src/recursion/
// {alu_base,alu_ext,select,poseidon2_wide,prefix_sum_checks}.rs and mod.rs; kernels under sp1-gpu/crates/sys): one H2D copy
// of the events, one kernel that writes the [width x height] trace. These are the MI355X counterparts, written against
// the CPU definitions of the traces:
//   BaseAlu           chips/alu_base.rs:L172-L208        row = BaseAluIo {out, in1, in2}
//   ExtAlu            chips/alu_ext.rs                    row = ExtAluIo<Block> {out[4], in1[4], in2[4]}
//   Select            chips/select.rs                     row = SelectIo {bit, out1, out2, in1, in2}
//   MemoryVar         chips/mem/variable.rs               row = VAR_EVENTS_PER_ROW (2) Blocks
//   PrefixSumChecks   chips/prefix_sum_checks.rs:L168-L221   x1, x2, acc, new_acc, field_acc, new_field_acc of the event
//   Poseidon2Wide     chips/poseidon2_wide/trace.rs:L44-L84 + `populate_perm`
//                     (/root/reference/crates/hypercube/src/operations/poseidon2/trace.rs:L29-L152): the states at the start
//                     of the 8 external rounds, the state entering the internal rounds, lane 0 after each of the first 19
//                     internal rounds, the output; padding rows are the permutation of the zero state
// (paths relative to /root/reference/crates/recursion/machine/src). Events are arrays of Montgomery words in the layout
// of the Rust `#[repr(C)]` event structs (/root/reference/crates/recursion/executor/src/lib.rs: BaseAluIo L87-L91,
// ExtAluIo L109-L113, SelectIo L230-L236, PrefixSumChecksEvent L272-L281, Poseidon2Event = {input[16], output[16]},
// MemEvent = Block). Output: column-major [width][height] tables, exactly what commit / LogUp-GKR / zerocheck consume —
// no row-major detour, no transpose. One lane per row: every column store of a wave is a 256 B run.
#include "device_ctx.hpp"
#include "poseidon2.hpp"

namespace sp1hip {
namespace {

// generic "the row is a sub-sequence of the event's words": map[c] = word of the event that goes to column c
template <int EVENT_WORDS, int WIDTH>
struct ColMap { int m[WIDTH]; };

template <int EVENT_WORDS, int WIDTH, int EVENTS_PER_ROW>
__global__ __launch_bounds__(256) void tracegen_copy_kernel(uint32_t* __restrict__ trace, uint64_t height,
                                                            const uint32_t* __restrict__ events, uint64_t n_events,
                                                            ColMap<EVENT_WORDS, WIDTH> map) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= height) return;
#pragma unroll
    for (int e = 0; e < EVENTS_PER_ROW; e++) {
        const uint64_t ev = r * EVENTS_PER_ROW + e;
        const bool real = ev < n_events;
#pragma unroll
        for (int c = 0; c < WIDTH; c++)
            trace[(size_t)(e * WIDTH + c) * height + r] = real ? events[ev * EVENT_WORDS + map.m[c]] : 0u;
    }
}

// Poseidon2Wide (degree 3): 179 columns per row
// (8 * 16 + 16 + 19 + 16 = 179 columns)
__global__ __launch_bounds__(256) void tracegen_poseidon2_wide_kernel(uint32_t* __restrict__ trace, uint64_t height,
                                                                     const uint32_t* __restrict__ events, uint64_t n_events,
                                                                     const p2::RoundConstants* __restrict__ rc) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= height) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = r < n_events ? events[r * 32 + i] : 0u;       // padding rows: populate_perm([0; 16])
    auto put = [&](int col, uint32_t v) { trace[(size_t)col * height + r] = v; };
    int col = 0;
#pragma unroll 1
    for (int round = 0; round < 8; round++) {
#pragma unroll
        for (int i = 0; i < 16; i++) put(col + i, s[i]);                                 // external_rounds_state[round]
        col += 16;
        if (round == 0) p2::external_linear(s);
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = p2::sbox(s[i], rc->ext[round][i]);
        p2::external_linear(s);
        if (round == 3) {
            // the internal rounds sit between external rounds 3 and 4; their columns come AFTER the 8 external states
#pragma unroll
            for (int i = 0; i < 16; i++) put(128 + i, s[i]);                             // internal_rounds_state
#pragma unroll 1
            for (int k = 0; k < 20; k++) {
                s[0] = p2::sbox(s[0], rc->internal[k]);
                p2::internal_linear_lazy(s);                                             // lanes 1..15 lazy, lane 0 canonical
                if (k < 19) put(144 + k, s[0]);                                          // internal_rounds_s0[k]
            }
#pragma unroll
            for (int i = 1; i < 16; i++) s[i] = kb::umin(s[i], s[i] - kb::P);            // lazy -> canonical
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) put(163 + i, s[i]);                                     // output_state
}

template <int EW, int W, int EPR>
int launch_copy(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events, const int (&m)[W], hipStream_t s) {
    SP1HIP_REQUIRE(d_trace && (d_events || n_events == 0), "null buffer");
    SP1HIP_REQUIRE((n_events + EPR - 1) / EPR <= height, "more events than rows");
    if (height == 0) return SP1HIP_SUCCESS;
    ColMap<EW, W> map;
    for (int c = 0; c < W; c++) map.m[c] = m[c];
    hipLaunchKernelGGL((tracegen_copy_kernel<EW, W, EPR>), dim3((unsigned)((height + 255) / 256)), dim3(256), 0, s, d_trace, height,
                       d_events, n_events, map);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

}  // namespace
}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_tracegen_recursion_base_alu(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                       sp1hip_stream_t stream) {
    const int m[3] = {0, 1, 2};
    return launch_copy<3, 3, 1>(d_trace, height, d_events, n_events, m, S(stream));
}

int sp1hip_tracegen_recursion_ext_alu(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                      sp1hip_stream_t stream) {
    const int m[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    return launch_copy<12, 12, 1>(d_trace, height, d_events, n_events, m, S(stream));
}

int sp1hip_tracegen_recursion_select(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                     sp1hip_stream_t stream) {
    const int m[5] = {0, 1, 2, 3, 4};
    return launch_copy<5, 5, 1>(d_trace, height, d_events, n_events, m, S(stream));
}

int sp1hip_tracegen_recursion_memory_var(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                         sp1hip_stream_t stream) {
    const int m[4] = {0, 1, 2, 3};
    return launch_copy<4, 4, 2>(d_trace, height, d_events, n_events, m, S(stream));
}

int sp1hip_tracegen_recursion_prefix_sum_checks(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                                sp1hip_stream_t stream) {
    // event: x1, x2[4], zero, one[4], acc[4], new_acc[4], field_acc, new_field_acc  ->  x1, x2, acc, new_acc, felt_acc, felt_new_acc
    const int m[15] = {0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19};
    return launch_copy<20, 15, 1>(d_trace, height, d_events, n_events, m, S(stream));
}

int sp1hip_tracegen_recursion_poseidon2_wide(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                             sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(d_trace && (d_events || n_events == 0), "null buffer");
    SP1HIP_REQUIRE(n_events <= height, "more events than rows");
    if (height == 0) return SP1HIP_SUCCESS;
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    hipLaunchKernelGGL(tracegen_poseidon2_wide_kernel, dim3((unsigned)((height + 255) / 256)), dim3(256), 0, S(stream), d_trace, height,
                       d_events, n_events, ctx->d_rc);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

}  // extern "C"
