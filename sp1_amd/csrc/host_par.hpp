// sp1_amd/csrc/host_par.hpp — fork/join over a few helper threads for the host arithmetic BETWEEN device hand-overs.
//
// A shard proof has stretches where the GPU waits for the host: the interaction-variable rounds of every LogUp-GKR layer
// (~11,000 extension-field products on <= 1024-entry tables, 0.35 ms per layer x 21 layers on one core: 7.4 of the ~20 ms
// the GPU idles inside a core-shaped proof). Those loops are embarrassingly parallel between two
// transcript steps, but each is only tens of microseconds long — too short for a thread pool that sleeps between
// jobs (a futex wake is ~50 us). So: helpers are woken once per `Scope` (one stage of one proof), SPIN between the jobs
// of that scope, and go back to sleep when the scope ends. One set of helpers per process: a second prover that is in
// flight on another stream finds them taken and simply runs its loops inline (`threads() == 1`).
// Field arithmetic is exact, so splitting a sum over threads and adding the parts in order gives the same words.
// SP1HIP_HOST_THREADS=<n> overrides the thread count (1 = never use helpers); default min(8, CPUs / (2 x processes of the
// node)), CPUs honouring the cgroup quota, processes = LOCAL_WORLD_SIZE (one prover process per GPU).
#pragma once
#include <cstddef>

namespace sp1hip {

class HostPar {
 public:
    class Scope {
     public:
        Scope();
        ~Scope();
        Scope(const Scope&) = delete;
        Scope& operator=(const Scope&) = delete;
        int threads() const { return threads_; }                      // parts a job is cut into (1 = inline)
        // f(part, begin, end) over `parts <= threads()` contiguous ranges of [0, n); part 0 runs on the caller; returns
        // when every part is done. Ranges shorter than min_per_part are not split further.
        template <class F>
        void run(size_t n, size_t min_per_part, F&& f) {
            int parts = threads_;
            if (min_per_part == 0) min_per_part = 1;
            if ((size_t)parts > n / min_per_part) parts = (int)(n / min_per_part);
            if (parts <= 1) { f(0, (size_t)0, n); return; }
            struct Ctx { F* f; size_t n; int parts; } ctx{&f, n, parts};
            dispatch(parts, [](void* c, int part) {
                Ctx* x = (Ctx*)c;
                const size_t b = x->n * (size_t)part / (size_t)x->parts, e = x->n * (size_t)(part + 1) / (size_t)x->parts;
                (*x->f)(part, b, e);
            }, &ctx);
        }
        static constexpr int MAX_THREADS = 16;
        // The helpers SPIN while the scope is awake. A stage that alternates long device phases with short bursts of host
        // loops parks them in between (they sleep on a condition variable) and wakes them a little BEFORE the next burst —
        // e.g. right before the launch whose result the burst consumes — so the ~50 us futex wake-up hides behind the device.
        void park();
        void wake();

     private:
        void dispatch(int parts, void (*fn)(void*, int), void* ctx);
        int threads_ = 1;
        bool owner_ = false;
        bool parked_ = false;
    };
};

}  // namespace sp1hip
