// sp1_amd/csrc/zc_jit.hpp — per-chip COMPILED zerocheck kernels next to the bytecode interpreter (zerocheck.hip).
//
// The interpreter spends two scalar / branch instructions of decode and dispatch per vector instruction of arithmetic
// (profiles/r02_pmc_zerocheck_chunks_sq.txt) and evaluates the three interpolation nodes of a row pair in three workgroups
// that each re-read the pair. For chips whose constraint program is small enough to compile in seconds — the RISC-V ALU /
// memory chips, the narrow recursion chips — the library generates straight-line HIP from the (scheduled, immediate-folded)
// SSA program, compiles it OFFLINE with hipcc (a background thread at first use, or ahead of time by
// `__graft_entry__.build()` into sp1_amd/lib/zc_cache/), caches the code object on disk by program hash and launches it
// for the large rounds; the interpreter keeps the small rounds, the wide chips (Poseidon2WideDeg3: 633 s to compile,
// DESIGN.md section 7) and everything whose kernel is not ready yet. Same sums either way, so the proof bytes do not depend
// on which path ran. The reference tiers its interpreter by register count instead
// (/root/reference/sp1-gpu/crates/sys/src/kernels.rs:L38-L117).
#pragma once
#include <atomic>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace sp1hip {

constexpr uint32_t ZC_JIT_VERSION = 4;            // bump when the generated code or zc_device.hpp / kb31.hpp change meaning
constexpr uint32_t ZC_JIT_MAX_INSTR = 600;        // longer programs stay interpreted: compile time grows super-linearly (793 instructions over
                                                  // 241 columns: 50 s) and the straight-line code starts to spill (331 / 781 VGPRs in base / extension form)
constexpr uint32_t ZC_JIT_MAX_WIDTH = 128;        // main + preprocessed columns
constexpr int ZC_JIT_EXT_MIN_BLOCKS = 2;             // __launch_bounds__(256, n) of the extension-field kernel: n workgroups per CU = n waves per SIMD
constexpr uint32_t ZC_JIT_MIN_TERMS = 1024;       // row pairs of a chip in a round from which its compiled kernel is used

struct ZcJitKernel;                               // one chip program's compiled kernel (process-wide, shared)

// HIP source of the two kernels (`zc_jit_first`: round 0 on base words, `zc_jit_ext`: later rounds) of one chip program.
// `ssa`: [n][3] scheduled SSA with the immediate forms and constraint indices of zerocheck.hip (ZcPlan::sched).
std::string zc_codegen(const uint32_t* ssa, uint32_t n, uint32_t main_w, uint32_t prep_w);

// hash of everything the code object depends on
uint64_t zc_jit_hash(const uint32_t* ssa, uint32_t n, uint32_t main_w, uint32_t prep_w);

// Look the kernel up (prebuilt directory next to the library, then the user cache) or queue its compilation. Never blocks
// on a compile. nullptr: not eligible (too long / too wide / SP1HIP_ZC_JIT=0).
std::shared_ptr<ZcJitKernel> zc_jit_request(const uint32_t* ssa, uint32_t n, uint32_t main_w, uint32_t prep_w);

// *fn = the kernel for the CURRENT device if the code object is ready (loads the module on first use), else nullptr.
int zc_jit_function(ZcJitKernel* k, bool first, hipFunction_t* fn);
// `want` side streams (created once per (device, caller stream)) and want + 1 events: [0] fork, [1 + k] join of stream k
int zc_jit_side_streams(hipStream_t main, int want, hipStream_t** streams, hipEvent_t** events);
bool zc_jit_enabled();                            // SP1HIP_ZC_JIT=1 (read per call; the compiled path is opt-in)
extern std::atomic<uint64_t> g_zc_jit_launches;

}  // namespace sp1hip
