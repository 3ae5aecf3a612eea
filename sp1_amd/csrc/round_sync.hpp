// sp1_amd/csrc/round_sync.hpp — hand a few words from the LAST workgroup of a kernel to the host without a
// second launch, a copy or a stream synchronise.
//
// Every sumcheck round of this library ends with "reduce the per-workgroup partial sums, give the host NS
// extension elements, let the host run the transcript". Done with a reduce kernel + hipMemcpyAsync +
// hipStreamSynchronize that is ~40 us of launch / copy / wake-up latency per round, and a shard proof has
// ~370 rounds. Here the workgroup that arrives last at an agent-scope counter reduces the partials and
// stores the sums plus a sequence number into mapped pinned host memory; the host spins on the sequence
// number (bounded).
//
// Inter-workgroup visibility WITHOUT fences (measured with bench/ubench_rs_finish.hip, profiles/r02_ubench_rs_finish.txt):
// the textbook protocol (plain stores, agent-scope release fence, counter; acquire fence, plain loads) makes every
// workgroup write back its XCD's whole L2 (`buffer_wbl2 sc1`) — in a fold round that is the freshly written half-size
// tables: 33 us instead of 16 for 730 workgroups with 64 KiB of output each, 176 instead of 60 for 4096. Only the
// 4 NS partial words per workgroup cross workgroups, so those are stored and loaded coherently (sc1: through to memory
// past the non-coherent L2s) and ordered by `s_waitcnt vmcnt(0)` + the workgroup barrier before the ticket; everything
// else a round writes becomes visible at the kernel boundary as usual. And the ticket is two-level (RS_GROUPS group
// counters on separate lines, then one): same-address agent-scope atomics serialise at ~17 ns each — 12 us for 730
// workgroups on a single counter.
#pragma once
#include <chrono>
#include <cstring>
#include <vector>
#include <algorithm>

#include "common.hpp"
#include "kb31.hpp"

// The fence-free hand-over below is correct only where (a) sc1 stores write through to memory past the XCD-private L2s
// and (b) `s_waitcnt vmcnt(0)` also waits for STORES (gfx942 / gfx950; gfx10+ counts stores in vscnt and would race
// silently). This library is written for gfx950 only, so any other device target is a build error, not a fallback.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "round_sync.hpp: the sc1 + vmcnt(0) hand-over is only valid on gfx942/gfx950"
#endif

namespace sp1hip {

struct RoundSync {                       // device-visible handles
    uint32_t* counter;                   // device word, zero between uses
    volatile uint32_t* host_slot;        // mapped pinned: [0] = sequence number, [1 ..] = the sums
};

constexpr uint32_t RS_GROUPS = 32, RS_GROUP_STRIDE = 64;             // counter words: group g at g * STRIDE, level 2 at GROUPS * STRIDE
constexpr size_t RS_COUNTER_BYTES = (size_t)(RS_GROUPS + 1) * RS_GROUP_STRIDE * 4;
constexpr size_t RS_SLOT_WORDS = 64;                                 // host slot: sequence number + up to 63 words of sums

// one lane per workgroup: true for the workgroup that arrives last; leaves every counter zero for the next launch
__device__ __forceinline__ bool rs_ticket_is_last(uint32_t* counter, uint32_t block_linear, uint32_t total_blocks) {
    const uint32_t g = block_linear % RS_GROUPS, members = (total_blocks - g + RS_GROUPS - 1) / RS_GROUPS;
    uint32_t* cg = counter + g * RS_GROUP_STRIDE;
    if (__hip_atomic_fetch_add(cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    __hip_atomic_store(cg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t* c2 = counter + RS_GROUPS * RS_GROUP_STRIDE;
    const uint32_t groups = total_blocks < RS_GROUPS ? total_blocks : RS_GROUPS;
    if (__hip_atomic_fetch_add(c2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != groups - 1) return false;
    __hip_atomic_store(c2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
// The same ticket for kernels whose workgroups write their PAYLOAD straight into the host slot (zc_reduce_kernel, the last fold
// of a LogUp-GKR layer) and let the last arrival publish the sequence number: here the workgroup that writes `seq` is not the
// one that wrote the payload, so the hand-over is a release by every producer and an acquire by the publisher in the HIP memory
// model — the ticket is an acq_rel read-modify-write and `seq` a system-scope release store. These launches have a
// few dozen workgroups per round, so the per-workgroup L2 write-back that rs_finish avoids (thousands of workgroups) costs
// nothing measurable here.
__device__ __forceinline__ bool rs_ticket_is_last_acq_rel(uint32_t* counter, uint32_t block_linear, uint32_t total_blocks) {
    const uint32_t g = block_linear % RS_GROUPS, members = (total_blocks - g + RS_GROUPS - 1) / RS_GROUPS;
    uint32_t* cg = counter + g * RS_GROUP_STRIDE;
    if (__hip_atomic_fetch_add(cg, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    __hip_atomic_store(cg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t* c2 = counter + RS_GROUPS * RS_GROUP_STRIDE;
    const uint32_t groups = total_blocks < RS_GROUPS ? total_blocks : RS_GROUPS;
    if (__hip_atomic_fetch_add(c2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != groups - 1) return false;
    __hip_atomic_store(c2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
__device__ __forceinline__ void rs_publish_seq(volatile uint32_t* slot, uint32_t seq) {
    __hip_atomic_store(const_cast<uint32_t*>(slot), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void rs_store_partial(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ kb::Ext rs_load_partial(const uint32_t* q) {
    kb::Ext e;
#pragma unroll
    for (int k = 0; k < 4; k++) e.c[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return e;
}

__device__ __forceinline__ uint32_t rs_wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, __shfl_down(v, off, 64));
    return v;
}

// Call from EVERY thread of EVERY workgroup of a 256-thread launch, once, at the end. acc: this thread's NS
// accumulators. partials: [total_blocks][4 NS] scratch. block_linear: this workgroup's index in [0, total_blocks).
template <int NS>
__device__ __forceinline__ void rs_finish(const kb::Ext (&acc)[NS], uint32_t* __restrict__ partials, uint32_t block_linear,
                                          uint32_t total_blocks, RoundSync rs, uint32_t seq) {
    __shared__ uint32_t sm[4][4 * NS];
    __shared__ uint32_t last_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = rs_wave_sum(acc[s].c[k]);
            if (lane == 0) sm[wave][4 * s + k] = w;
        }
    __syncthreads();
    if (threadIdx.x < 4 * NS) {
        uint32_t a = 0;
        for (int i = 0; i < 4; i++) a = kb::add(a, sm[i][threadIdx.x]);
        rs_store_partial(&partials[(size_t)block_linear * 4 * NS + threadIdx.x], a);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_flag = rs_ticket_is_last(rs.counter, block_linear, total_blocks);
    __syncthreads();
    if (!last_flag) return;
    // the last workgroup: total over all partials
    kb::Ext tot[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) tot[s] = kb::ext_zero();
    for (uint32_t i = threadIdx.x; i < total_blocks; i += 256)
#pragma unroll
        for (int s = 0; s < NS; s++)
            tot[s] = kb::ext_add(tot[s], rs_load_partial(partials + ((size_t)i * NS + s) * 4));
    __syncthreads();                       // sm is reused
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = rs_wave_sum(tot[s].c[k]);
            if (lane == 0) sm[wave][4 * s + k] = w;
        }
    __syncthreads();
    if (threadIdx.x < 4 * NS) {
        uint32_t a = 0;
        for (int i = 0; i < 4; i++) a = kb::add(a, sm[i][threadIdx.x]);
        rs.host_slot[1 + threadIdx.x] = a;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the slot is uncached host memory: acknowledged stores are ordered before seq
    __syncthreads();
    if (threadIdx.x == 0) rs.host_slot[0] = seq;
}

// Host side: owns the counter word and the pinned slot; wait() spins until the kernel that was given `seq` has
// published (bounded; also notices a failed launch). The pair (device counter, pinned slot) comes from a process-wide
// free list: hipMalloc / hipFree / hipHostMalloc synchronise the device and would serialise provers that run
// concurrently on other streams.
struct RoundSyncSlot { uint32_t* d_counter; uint32_t* h_slot; };
int round_sync_acquire(RoundSyncSlot* out, hipStream_t stream);        // runtime.hip: a NEW slot is zeroed on `stream`, ahead of its first user
void round_sync_release(RoundSyncSlot slot);

struct RoundSyncHost {
    uint32_t* d_counter = nullptr;
    uint32_t* h_slot = nullptr;            // pinned + mapped, RS_SLOT_WORDS words
    uint32_t seq = 0;
    hipStream_t s = nullptr;
    int init(hipStream_t stream) {
        s = stream;
        RoundSyncSlot slot;
        SP1HIP_TRY(round_sync_acquire(&slot, stream));
        d_counter = slot.d_counter;
        h_slot = slot.h_slot;
        seq = h_slot[0];                   // continue the slot's sequence (its counter is zero between uses)
        return SP1HIP_SUCCESS;
    }
    bool pending = false;                  // a kernel that will write this slot may still be in flight
    ~RoundSyncHost() {
        // an early exit (error return, timeout) can leave a round kernel queued that still increments the counter and
        // writes the sequence number: drain the stream before another prover may acquire the slot
        if (d_counter && pending) (void)hipStreamSynchronize(s);
        if (d_counter) round_sync_release(RoundSyncSlot{d_counter, h_slot});
    }
    RoundSync next() { seq++; pending = true; return RoundSync{d_counter, (volatile uint32_t*)h_slot}; }
    // a launch that takes tickets from the counter but finishes its round on the device (no host slot): still "pending"
    // until the caller has seen a later hand-over on the stream (settled())
    RoundSync chained() { pending = true; return RoundSync{d_counter, nullptr}; }
    void settled() { pending = false; }
    // copies n_words sums (from slot[1..]) into out
    int wait(uint32_t* out, int n_words) { return wait_for(seq, out, n_words); }
    // the same for the hand-over numbered `which` (a launch armed behind a HostGate may already have taken the next number)
    int wait_for(uint32_t which, uint32_t* out, int n_words) {
        volatile uint32_t* slot = h_slot;
        SP1HIP_TRY(wait_for_seq(slot, which, s, "a sumcheck round result"));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        pending = false;
        for (int k = 0; k < n_words; k++) out[k] = slot[1 + k];
        return SP1HIP_SUCCESS;
    }
};

// ---- HostGate: hand a few words from the HOST to a launch that is already enqueued.
// Every sumcheck round ends with the host sampling a challenge the next launch needs. Launching at that moment puts the launch
// on the critical path: ~12 us from the host's store to the completion of a one-wave kernel, of which ~6 us are the dispatch of a
// kernel behind another one. So a small pass is enqueued one hand-over EARLY — its dispatch overlaps the host's half of the round
// trip — and lane 0 of every workgroup polls a ticket word in mapped pinned memory until the host has written the challenges
// into the ticket's slot (`open()`: the slot first, then the ticket with release order). hipStreamWaitValue32 was tried for the
// wait: ROCm 7.2 implements it as a one-workgroup polling KERNEL between the two launches, i.e. two dependent dispatches instead
// of one, and nothing is gained (bench/ubench/ubench_wait_value.hip; profiles/r06_gkr_launch_trace_stream_wait_value.txt).
// Tickets only grow; the poll is bounded by the wall clock (GATE_TIMEOUT_TICKS) and a gate that goes out of scope with armed
// tickets opens them all, so neither an error return nor a dead host thread leaves a kernel spinning for ever.
constexpr uint32_t GATE_RING = 64;                     // slots of 8 extension elements each; at most a few tickets are armed at a time
constexpr uint32_t GATE_SLOT_WORDS = 32, GATE_SLOT0 = 64;
constexpr uint64_t GATE_TIMEOUT_TICKS = 300000000ull;  // 3 s of the 100 MHz wall clock
struct HostGateBlock { uint32_t* h; };                 // [0] = the ticket word (its own cache line), slots from word GATE_SLOT0
int host_gate_acquire(HostGateBlock* out);             // runtime.hip (process-wide free list per device, like the mailbox slots)
void host_gate_release(HostGateBlock b);

struct GateArg { const uint32_t* block; uint32_t ticket; };   // kernel argument: block == nullptr: not gated

struct HostGate {
    uint32_t* h = nullptr;
    uint32_t issued = 0, opened = 0;                   // tickets armed / opened so far (both continue the block's sequence)
    hipStream_t s = nullptr;
    int init(hipStream_t stream) {
        s = stream;
        HostGateBlock b;
        SP1HIP_TRY(host_gate_acquire(&b));
        h = b.h;
        issued = opened = h[0];
        return SP1HIP_SUCCESS;
    }
    ~HostGate() {
        if (!h) return;
        if (opened != issued) {                        // an early exit: let the armed launches run (on whatever the slots hold) and drain
            __atomic_store_n(h, issued, __ATOMIC_RELEASE);
            (void)hipStreamSynchronize(s);
        }
        host_gate_release(HostGateBlock{h});
    }
    HostGate() = default;
    HostGate(const HostGate&) = delete;
    HostGate& operator=(const HostGate&) = delete;
    GateArg arm() { issued++; return GateArg{h, issued}; }      // the launch that takes this argument waits for open(ticket)
    void open(uint32_t ticket, const uint32_t* words, int n_words) {
        uint32_t* slot = h + GATE_SLOT0 + (size_t)(ticket % GATE_RING) * GATE_SLOT_WORDS;
        for (int k = 0; k < n_words; k++) slot[k] = words[k];
        __atomic_store_n(h, ticket, __ATOMIC_RELEASE);
        opened = ticket;
    }
};
// device side, all threads of the workgroup: wait for the ticket, then the first n_words (<= 32) of its slot through LDS — ONE
// polling lane and one read of the slot per workgroup (every lane reading mapped host memory itself is thousands of uncached reads
// across PCIe per workgroup: measured +85 us per small pass)
__device__ __forceinline__ void gate_wait_load(const GateArg& g, uint32_t* lds_words, int n_words) {
    if (threadIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(g.block, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - g.ticket) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > GATE_TIMEOUT_TICKS) break;
        }
    }
    __syncthreads();
    const uint32_t* slot = g.block + GATE_SLOT0 + (size_t)(g.ticket % GATE_RING) * GATE_SLOT_WORDS;
    if ((int)threadIdx.x < n_words) lds_words[threadIdx.x] = __hip_atomic_load(slot + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
}

// ---- Mailbox: "copy these few device words to the host and wait for them" without hipMemcpyAsync + hipStreamSynchronize.
// A one-workgroup kernel, ordered on the stream behind the producer, stores the words into mapped pinned memory and
// then a sequence number; the host spins on the sequence number. Seeing it also means everything enqueued on the
// stream before the fetch has completed (in-order stream), so host staging buffers handed to earlier
// hipMemcpyAsync calls may be reused. Measured against the copy + synchronise pair this saves ~30-50 us per
// round trip (no blit dispatch on completion, no interrupt wake-up); a shard proof has ~170 of them outside GKR.
constexpr uint32_t MAILBOX_WORDS = 16384 + 32;  // payload capacity (64 KiB + room for a 16-byte aligned start, wait_next)

int mailbox_publish(const uint32_t* d_src, uint32_t n_words, uint32_t* h_slot, uint32_t seq, hipStream_t s);   // runtime.hip

struct MailboxSlot { uint32_t* h_slot; };       // [0] = sequence number, [1 .. MAILBOX_WORDS] payload
int mailbox_acquire(MailboxSlot* out);          // runtime.hip (process-wide free list, like the round-sync slots)
void mailbox_release(MailboxSlot slot);

struct Mailbox {
    uint32_t* h_slot = nullptr;
    uint32_t seq = 0;
    hipStream_t s = nullptr;
    int init(hipStream_t stream) {
        s = stream;
        MailboxSlot slot;
        SP1HIP_TRY(mailbox_acquire(&slot));
        h_slot = slot.h_slot;
        seq = h_slot[0];
        return SP1HIP_SUCCESS;
    }
    bool pending = false;
    ~Mailbox() {
        if (h_slot && pending) (void)hipStreamSynchronize(s);      // see ~RoundSyncHost
        if (h_slot) mailbox_release(MailboxSlot{h_slot});
    }
    // d_src[0 .. n_words) -> out, after everything already enqueued on the stream. n_words == 0: a pure fence.
    int fetch(const void* d_src, size_t n_words, void* out) {
        if (n_words > MAILBOX_WORDS) {           // too large for the slot: the classic pair
            SP1HIP_HIP(hipMemcpyAsync(out, d_src, n_words * 4, hipMemcpyDeviceToHost, s));
            SP1HIP_HIP(hipStreamSynchronize(s));
            return SP1HIP_SUCCESS;
        }
        SP1HIP_TRY(mailbox_publish((const uint32_t*)d_src, (uint32_t)n_words, h_slot, seq + 1, s));
        return wait_next(out, n_words);
    }
    // For kernels that write the slot themselves (payload words [1 ..], then `seq + 1` into word 0): waits for
    // that sequence number and copies the payload out.
    // first_word: where the kernel put the payload (4 for kernels that store 16-byte vectors: word 1 is not aligned).
    int wait_next(void* out, size_t n_words, size_t first_word = 1) {
        seq++;
        pending = true;
        volatile uint32_t* slot = h_slot;
        SP1HIP_TRY(wait_for_seq(slot, seq, s, "a device result"));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        pending = false;
        uint32_t* o = (uint32_t*)out;
        for (size_t k = 0; k < n_words; k++) o[k] = slot[first_word + k];
        return SP1HIP_SUCCESS;
    }
};

// ---- PinnedStage: small host -> device uploads without the synchronous bounce of pageable hipMemcpyAsync.
// hipMemcpyAsync from ordinary host memory stages the bytes inside the call (~35-45 us each, measured in the
// kernel trace of a shard proof: ~190 descriptor / table uploads cost ~6 ms of host time). Here the bytes are
// memcpy'd into a pinned block (bump allocation, never reused within one prover call) and the copy engine reads
// them from there: the call returns in a few microseconds. A call that outgrows the block falls back to the
// pageable path. Release the stage only after a hand-over that orders the host behind the stream (Mailbox::fetch).
constexpr size_t PINNED_STAGE_BYTES = (size_t)2 << 20;
struct PinnedBlock { uint8_t* h; };
int pinned_stage_acquire(PinnedBlock* out);     // runtime.hip
void pinned_stage_release(PinnedBlock b);

struct PinnedStage {
    uint8_t* h = nullptr;
    size_t used = 0;
    hipStream_t s = nullptr;
    std::vector<uint8_t*> extra;           // whole blocks taken by uploads that do not fit what is left of `h`
    int init(hipStream_t stream) {
        s = stream;
        PinnedBlock b;
        SP1HIP_TRY(pinned_stage_acquire(&b));
        h = b.h;
        return SP1HIP_SUCCESS;
    }
    ~PinnedStage() {
        // a DMA that still reads the block must finish before the block is handed to another prover; on the normal path
        // the stream is already idle here (the caller has just received its last result), so the query is all it costs
        if (h && used && hipStreamQuery(s) != hipSuccess) (void)hipStreamSynchronize(s);
        if (h) pinned_stage_release(PinnedBlock{h});
        for (uint8_t* b : extra) pinned_stage_release(PinnedBlock{b});
    }
    int upload(void* d_dst, const void* src, size_t bytes) {
        if (bytes == 0) return SP1HIP_SUCCESS;
        const size_t at = (used + 63) & ~(size_t)63;
        if (h && at + bytes <= PINNED_STAGE_BYTES) {
            memcpy(h + at, src, bytes);
            used = at + bytes;
            SP1HIP_HIP(hipMemcpyAsync(d_dst, h + at, bytes, hipMemcpyHostToDevice, s));
        } else if (h && bytes <= 16 * PINNED_STAGE_BYTES) {
            // a large table (the LogUp-GKR pass descriptors of a core shard are ~10 MB): pieces through blocks of their own.
            // From pageable memory the same copy is staged INSIDE the call at ~3 GB/s (3.7 ms of host time measured).
            size_t off = 0;
            while (off < bytes) {
                PinnedBlock b;
                if (pinned_stage_acquire(&b) != SP1HIP_SUCCESS) break;
                extra.push_back(b.h);
                const size_t n = std::min(PINNED_STAGE_BYTES, bytes - off);
                memcpy(b.h, (const uint8_t*)src + off, n);
                SP1HIP_HIP(hipMemcpyAsync((uint8_t*)d_dst + off, b.h, n, hipMemcpyHostToDevice, s));
                off += n;
            }
            used = std::max<size_t>(used, 1);
            if (off < bytes) SP1HIP_HIP(hipMemcpyAsync((uint8_t*)d_dst + off, (const uint8_t*)src + off, bytes - off, hipMemcpyHostToDevice, s));
        } else {
            SP1HIP_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, s));
        }
        return SP1HIP_SUCCESS;
    }
};

}  // namespace sp1hip
