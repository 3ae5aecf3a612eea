// sp1_amd/csrc/host_par.hip — the helper threads behind host_par.hpp (host code only).
#include "host_par.hpp"

#include <atomic>
#include <condition_variable>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>

#include <pthread.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define SP1HIP_CPU_PAUSE() _mm_pause()
#else
#define SP1HIP_CPU_PAUSE() std::this_thread::yield()
#endif

namespace sp1hip {
namespace {

struct Pool {
    int helpers = 0;                                   // threads besides the caller
    std::mutex m;
    std::condition_variable cv;
    bool active = false;                               // guarded by m for the sleepers, mirrored in `spinning`
    std::atomic<bool> spinning{false};
    std::atomic<bool> taken{false};                    // a Scope owns the helpers
    // the current job: written by the owner before `epoch` is bumped (release), read by helpers after they see it (acquire)
    void (*fn)(void*, int) = nullptr;
    void* ctx = nullptr;
    int parts = 0;
    alignas(64) std::atomic<uint64_t> epoch{0};
    alignas(64) std::atomic<int> acked{0};             // helpers that are through with the current epoch (all of them ack)

    void helper(int index) {                           // index 1 .. helpers
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return active; });
            }
            uint32_t idle = 0;
            while (spinning.load(std::memory_order_acquire)) {
                const uint64_t e = epoch.load(std::memory_order_acquire);
                // (a helper that has spun for a while without work gives its time slice away: on an oversubscribed host the
                // thread it is waiting for may not be running)
                if (e == seen) { if (++idle > 4096) { idle = 0; std::this_thread::yield(); } else SP1HIP_CPU_PAUSE(); continue; }
                idle = 0;
                seen = e;
                if (index < parts) fn(ctx, index);
                acked.fetch_add(1, std::memory_order_release);
            }
        }
    }
};

Pool* g_pool_for_fork = nullptr;

Pool* pool() {
    // never destroyed: the helpers are detached and sleep on the condition variable when no scope is open
    static Pool* p = [] {
        Pool* q = new Pool();
        int n;
        if (const char* e = getenv("SP1HIP_HOST_THREADS")) n = atoi(e);
        else {
            unsigned cpus = std::max(1u, std::thread::hardware_concurrency());
            // a container's CPU quota (cgroup v2 "quota period"): spinning helpers beyond it would only get throttled
            if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
                long long quota = 0, period = 0;
                if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                    cpus = std::min<unsigned>(cpus, (unsigned)std::max<long long>(1, quota / period));
                fclose(f);
            }
            // one prover process per GPU (torchrun exports LOCAL_WORLD_SIZE): the processes of a node share the CPUs, and every
            // prover also has a thread that polls for device hand-overs
            unsigned procs = 1;
            if (const char* lw = getenv("LOCAL_WORLD_SIZE")) procs = (unsigned)std::max(1, atoi(lw));
            n = (int)std::min<unsigned>(8u, std::max(1u, cpus / (2 * procs)));
        }
        n = std::max(1, std::min(n, HostPar::Scope::MAX_THREADS));
        q->helpers = n - 1;
        for (int i = 1; i <= q->helpers; i++) std::thread([q, i] { q->helper(i); }).detach();
        g_pool_for_fork = q;
        // threads do not survive fork(): a child that inherited `helpers > 0` would wait for acknowledgements forever
        (void)pthread_atfork(nullptr, nullptr, [] {
            if (Pool* c = g_pool_for_fork) { c->helpers = 0; c->taken.store(false, std::memory_order_relaxed); }
        });
        return q;
    }();
    return p;
}

}  // namespace

HostPar::Scope::Scope() {
    Pool* p = pool();
    if (p->helpers == 0) return;
    bool expected = false;
    if (!p->taken.compare_exchange_strong(expected, true, std::memory_order_acq_rel)) return;       // another prover has them
    owner_ = true;
    threads_ = p->helpers + 1;
    p->spinning.store(true, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->active = true;
    }
    p->cv.notify_all();
}

HostPar::Scope::~Scope() {
    if (!owner_) return;
    park();
    pool()->taken.store(false, std::memory_order_release);
}

void HostPar::Scope::park() {
    if (!owner_ || parked_) return;
    Pool* p = pool();
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->active = false;
    }
    p->spinning.store(false, std::memory_order_release);
    parked_ = true;
}

void HostPar::Scope::wake() {
    if (!owner_ || !parked_) return;
    Pool* p = pool();
    p->spinning.store(true, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->active = true;
    }
    p->cv.notify_all();
    parked_ = false;
}

void HostPar::Scope::dispatch(int parts, void (*fn)(void*, int), void* ctx) {
    Pool* p = pool();
    if (parked_) wake();                               // a job while parked: correct, only pays the wake-up here
    p->fn = fn; p->ctx = ctx; p->parts = parts;
    p->acked.store(0, std::memory_order_relaxed);
    p->epoch.fetch_add(1, std::memory_order_release);
    fn(ctx, 0);
    // every helper acknowledges every epoch (also those without a part): none can be between "saw the epoch" and "read
    // the job" when the next job is written
    uint32_t idle = 0;
    while (p->acked.load(std::memory_order_acquire) != p->helpers) {
        if (++idle > 4096) { idle = 0; std::this_thread::yield(); } else SP1HIP_CPU_PAUSE();
    }
}

}  // namespace sp1hip

extern "C" int sp1hip_host_threads(void) { return sp1hip::pool()->helpers + 1; }
