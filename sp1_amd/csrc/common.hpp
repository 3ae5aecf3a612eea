// sp1_amd/csrc/common.hpp — shared host-side plumbing for libsp1hip.so: status codes, the
// thread-local error message, HIP error mapping, launch helpers, per-device constant upload.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/sp1hip.h"

namespace sp1hip {

// Pointers that reach a kernel INSIDE a descriptor / pointer table read from memory are generic pointers to the compiler,
// which then emits FLAT loads and stores (address-space check per lane, both memory counters, no scalar base). gptr()
// restores the global address space at the access site; q4 is the 16-byte word extension elements travel as.
typedef uint32_t q4_t __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ const __attribute__((address_space(1))) T* gptr(const T* p) {
    return (const __attribute__((address_space(1))) T*)p;
}
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T* gptr(T* p) {
    return (__attribute__((address_space(1))) T*)p;
}

void set_error(const char* fmt, ...);
// what the calling thread is in the middle of (a stage of a shard proof): carried by the time-out messages of the device -> host
// hand-overs, whose wait loops know nothing about their caller. wait_timeout_seconds(): SP1HIP_WAIT_TIMEOUT_S, default 60.
void set_stage_note(const char* note);
const char* stage_note();
int wait_timeout_seconds();
// Device -> host hand-overs (round_sync.hpp) wait on a sequence number in mapped pinned memory. wait_for_seq() spins — unless the
// calling thread is inside a WaitPlan (one whole shard proof) that has seen this hand-over before: a proof is a fixed sequence of
// ~700 hand-overs whose durations repeat from proof to proof of the same shape, so the thread SLEEPS until shortly before the
// expected arrival (clock_nanosleep, absolute) and spins only through the margin. A hand-over that arrives earlier than
// predicted is found late by at most what was left of the sleep, and the next prediction is shortened; unknown shapes, the first
// proof of a shape and hand-overs under 150 us are spun as before. SP1HIP_WAIT=spin turns the sleeping off.
int wait_for_seq(volatile uint32_t* slot, uint32_t seq, hipStream_t s, const char* what);
struct WaitPlan {                                   // RAII: opened by the shard prover around one proof
    bool opened;
    bool ok = false;                                // set by the owner when the proof succeeded: only then is the timeline kept
    explicit WaitPlan(uint64_t shape_signature);
    ~WaitPlan();
    WaitPlan(const WaitPlan&) = delete;
    WaitPlan& operator=(const WaitPlan&) = delete;
};
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) instead of once per launch (a driver call in
// front of every zerocheck round launch and every 256-point encode pass).
int ensure_dynamic_lds(const void* kernel, int bytes);
// Provers (shard proofs, or stand-alone stage calls) in flight in this process, counted once per calling thread however the
// entry points nest. The zerocheck spreads a round's launches over fork streams only while it is the ONLY prover: with several
// provers in flight the device is filled by the other proofs anyway, and the forks' cross-stream events on the process's four
// hardware queues cost more than they hide (3-slot pool on the real-chip shard: 55.1 ms per proof without forks, 60.6 with).
struct ActiveProver {
    bool counted;
    ActiveProver();
    ~ActiveProver();
    ActiveProver(const ActiveProver&) = delete;
    ActiveProver& operator=(const ActiveProver&) = delete;
};
int active_provers();
int map_hip_error(hipError_t e, const char* what);

inline hipStream_t S(sp1hip_stream_t s) { return static_cast<hipStream_t>(s); }

// Buffer arena (runtime.hip): stream-keyed free lists; alloc never blocks once the working set is cached.
int arena_alloc(void** ptr, size_t bytes, hipStream_t stream);
void arena_free(void* ptr, size_t bytes, hipStream_t stream);
size_t arena_trim();
void arena_miss_stats(uint64_t* n, uint64_t* bytes);   // requests that went to hipMalloc since the process started
size_t arena_release_stream(hipStream_t stream);      // blocks cached for a stream that is going away

// Optional event bracketing of a kernel launch (see sp1hip_timers_* in include/sp1hip.h).
bool timers_on();
int timer_begin(const char* name, hipStream_t s);      // index of the record, -1 if none
void timer_end(int idx, hipStream_t s);
struct ScopedTimer {
    hipStream_t s;
    int idx;
    ScopedTimer(const char* name, hipStream_t stream) : s(stream), idx(timers_on() ? timer_begin(name, stream) : -1) {}
    ~ScopedTimer() { if (idx >= 0) timer_end(idx, s); }
};

// roctx ranges around the stages of a proof (the reference's NvtxLayer, sp1-gpu/crates/tracing/src/tracer.rs): what
// `rocprofv3 --marker-trace` shows next to the kernel trace. The marker library (librocprofiler-sdk-roctx, else libroctx64) is
// looked up with dlopen the first time a range opens; without it — or with SP1HIP_ROCTX=0 — a range costs one branch.
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char* name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// A side stream (and n_events reusable events) paired with the caller's stream (runtime.hip).
int aux_stream_for(hipStream_t main, int n_events, hipStream_t* aux, hipEvent_t** events);
int fork_streams_for(hipStream_t main, int n, hipStream_t** streams, hipEvent_t** events);
void release_stream_helpers(hipStream_t main);

}  // namespace sp1hip

#define SP1HIP_TRY(expr)                                                   \
    do {                                                                   \
        int _st = (expr);                                                  \
        if (_st != SP1HIP_SUCCESS) return _st;                             \
    } while (0)

#define SP1HIP_HIP(expr)                                                   \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) return sp1hip::map_hip_error(_e, #expr);     \
    } while (0)

#define SP1HIP_REQUIRE(cond, msg)                                          \
    do {                                                                   \
        if (!(cond)) {                                                     \
            sp1hip::set_error("%s: %s", __func__, msg);                    \
            return SP1HIP_ERROR_INVALID_ARGUMENT;                          \
        }                                                                  \
    } while (0)

#define SP1HIP_LAUNCH_CHECK() SP1HIP_HIP(hipGetLastError())
