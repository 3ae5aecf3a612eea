// sp1_amd/csrc/runtime.hip — runtime half of the C ABI (memory, streams, events, errors) and the
// per-device context. HIP-native equivalents of the reference's runtime shims
// (/root/reference/sp1-gpu/crates/sys/src/runtime.rs:L16-L172): hipMallocAsync-backed allocation,
// streams, events; no globals other than the per-device contexts.
#include <dlfcn.h>
#include <sys/prctl.h>
#include <time.h>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "device_ctx.hpp"
#include "round_sync.hpp"

namespace sp1hip {

static thread_local std::string g_last_error;
static thread_local const char* g_stage_note = "";
void set_stage_note(const char* note) { g_stage_note = note ? note : ""; }
const char* stage_note() { return g_stage_note; }
static std::atomic<int> g_active_provers{0};
static thread_local int g_prover_depth = 0;
ActiveProver::ActiveProver() : counted(g_prover_depth++ == 0) { if (counted) g_active_provers.fetch_add(1, std::memory_order_relaxed); }
ActiveProver::~ActiveProver() { g_prover_depth--; if (counted) g_active_provers.fetch_sub(1, std::memory_order_relaxed); }
int active_provers() { return g_active_provers.load(std::memory_order_relaxed); }
int ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex m;
    static std::map<std::pair<int, const void*>, int> done;       // (device, kernel) -> the size already granted
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(m);
        auto it = done.find({dev, kernel});
        if (it != done.end() && it->second >= bytes) return SP1HIP_SUCCESS;
    }
    SP1HIP_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    std::lock_guard<std::mutex> lock(m);
    int& v = done[{dev, kernel}];
    v = std::max(v, bytes);
    return SP1HIP_SUCCESS;
}
int wait_timeout_seconds() {
    static const int t = [] { const char* e = getenv("SP1HIP_WAIT_TIMEOUT_S"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 60; }();
    return t;
}

// ---- predictive hand-over waits (common.hpp: wait_for_seq / WaitPlan)
namespace {
struct WaitTimeline { std::vector<float> last, prev; uint64_t used = 0; };       // microseconds per hand-over ordinal
struct WaitState {
    std::map<uint64_t, WaitTimeline> shapes;          // by shape signature (a thread proves a handful of shapes)
    WaitTimeline* cur = nullptr;
    uint64_t cur_sig = 0;
    std::vector<float> rec;                           // this proof's measured waits
    int depth = 0;
    uint64_t clock = 0;
    bool slack_set = false;
};
thread_local WaitState g_wait;
// (measured, round 5: with a 60 us margin the step was 0.6-1.3 % slower than pure spinning — wake-ups from an idle state — so
// the margin is 100 us; the caller thread then burns ~30 ms of CPU per 80 ms proof instead of all of it)
constexpr float WAIT_MIN_PREDICTED_US = 250.f, WAIT_MARGIN_US = 100.f, WAIT_MARGIN_FRAC = 0.10f;
bool wait_sleeping_enabled() {
    static const bool on = [] { const char* e = getenv("SP1HIP_WAIT"); return !(e && strcmp(e, "spin") == 0); }();
    return on;
}
}  // namespace

WaitPlan::WaitPlan(uint64_t sig) : opened(false) {
    WaitState& w = g_wait;
    if (w.depth++ > 0 || !wait_sleeping_enabled()) return;
    opened = true;
    if (!w.slack_set) { (void)prctl(PR_SET_TIMERSLACK, 2000UL, 0UL, 0UL, 0UL); w.slack_set = true; }   // 2 us instead of the default 50
    auto it = w.shapes.find(sig);
    w.cur = it == w.shapes.end() ? nullptr : &it->second;
    w.cur_sig = sig;
    w.rec.clear();
    w.rec.reserve(w.cur ? w.cur->last.size() + 16 : 1024);
}

WaitPlan::~WaitPlan() {
    WaitState& w = g_wait;
    w.depth--;
    if (!opened) return;
    if (ok && !w.rec.empty()) {
        if (!w.cur && w.shapes.size() >= 8) {          // keep the eight most recently used shapes
            auto victim = w.shapes.begin();
            for (auto it = w.shapes.begin(); it != w.shapes.end(); ++it) if (it->second.used < victim->second.used) victim = it;
            w.shapes.erase(victim);
        }
        WaitTimeline& t = w.shapes[w.cur_sig];
        t.used = ++w.clock;
        if (t.last.size() == w.rec.size()) t.prev.swap(t.last); else t.prev = w.rec;     // a different hand-over count: start over
        t.last = w.rec;
    }
    w.cur = nullptr;
    w.rec.clear();
}

int wait_for_seq(volatile uint32_t* slot, uint32_t seq, hipStream_t s, const char* what) {
    using clk = std::chrono::steady_clock;
    WaitState& w = g_wait;
    const bool planned = w.depth > 0 && wait_sleeping_enabled();
    const auto t0 = clk::now();
    bool overslept = false;
    float predicted = 0.f;
    if (planned && w.cur && slot[0] != seq) {
        const size_t ord = w.rec.size();
        if (ord < w.cur->last.size() && ord < w.cur->prev.size()) {
            predicted = std::min(w.cur->last[ord], w.cur->prev[ord]);
            if (predicted >= WAIT_MIN_PREDICTED_US) {
                const float sleep_us = predicted - std::max(WAIT_MARGIN_US, WAIT_MARGIN_FRAC * predicted);
                timespec now_ts;
                clock_gettime(CLOCK_MONOTONIC, &now_ts);
                const long long until = (long long)now_ts.tv_sec * 1000000000LL + now_ts.tv_nsec + (long long)(sleep_us * 1e3f);
                timespec ts{(time_t)(until / 1000000000LL), (long)(until % 1000000000LL)};
                while (clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, nullptr) == EINTR) {}
                overslept = slot[0] == seq;
            }
        }
    }
    uint64_t spins = 0;
    while (slot[0] != seq) {
        if ((++spins & 0xffff) == 0) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) { set_error("kernel failed while the host waited for %s", what); return map_hip_error(q, "kernel failed while the host waited for a device result"); }
            if (clk::now() - t0 > std::chrono::seconds(wait_timeout_seconds())) {
                set_error("timed out waiting for %s (stage %s; expected sequence %u, the slot holds %u; stream query %d)",
                          what, stage_note(), seq, slot[0], (int)q);
                return SP1HIP_ERROR_RUNTIME;
            }
        }
    }
    if (planned) {
        float us = std::chrono::duration<float, std::micro>(clk::now() - t0).count();
        // the result was already there when the sleep ended: its true arrival is unknown, so wake earlier next time
        if (overslept) us = 0.75f * std::min(us, predicted);
        w.rec.push_back(us);
    }
    return SP1HIP_SUCCESS;
}

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int map_hip_error(hipError_t e, const char* what) {
    set_error("%s failed: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();  // clear the sticky error
    switch (e) {
        case hipErrorOutOfMemory: return SP1HIP_ERROR_OUT_OF_MEMORY;
        case hipErrorNotReady: return SP1HIP_ERROR_NOT_READY;
        case hipErrorNoDevice:
        case hipErrorInvalidDevice: return SP1HIP_ERROR_NO_DEVICE;
        case hipErrorInvalidValue: return SP1HIP_ERROR_INVALID_ARGUMENT;
        default: return SP1HIP_ERROR_RUNTIME;
    }
}

struct TimerRec { std::string name; hipEvent_t a, b; };
static std::mutex g_timer_mutex;
static bool g_timers_on = false;
static std::vector<TimerRec> g_timer_recs;

bool timers_on() { return g_timers_on; }
int timer_begin(const char* name, hipStream_t s) {
    TimerRec r;
    r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    (void)hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> lock(g_timer_mutex);
    g_timer_recs.push_back(r);
    return (int)g_timer_recs.size() - 1;
}
void timer_end(int idx, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_timer_mutex);
    if (idx < 0 || idx >= (int)g_timer_recs.size()) return;   // reset in between: drop the sample
    (void)hipEventRecord(g_timer_recs[idx].b, s);
}

// ---- helper streams ------------------------------------------------------------------------------
// One high-priority side stream per (device, caller stream), created on first use and kept: commit_mles
// runs the Reed-Solomon encodes there while the caller's stream hashes the batches already encoded.
struct AuxRec { hipStream_t aux; std::vector<hipEvent_t> events; };
static std::mutex g_aux_mutex;
static std::map<std::pair<int, hipStream_t>, AuxRec> g_aux;

int aux_stream_for(hipStream_t main, int n_events, hipStream_t* aux, hipEvent_t** events) {
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_aux_mutex);
    AuxRec& r = g_aux[{dev, main}];
    if (!r.aux) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
        SP1HIP_HIP(hipStreamCreateWithPriority(&r.aux, hipStreamNonBlocking, hi));
    }
    while ((int)r.events.size() < n_events) {
        hipEvent_t e;
        SP1HIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        r.events.push_back(e);
    }
    *aux = r.aux;
    *events = r.events.data();   // stable until the next call for this stream (one host thread per stream)
    return SP1HIP_SUCCESS;
}

// Fork streams: `n` ordinary-priority streams per (device, caller stream) for launches of one round that do not depend on
// each other (the zerocheck's interpreter groups and fused pieces). events[0] is the fork event, events[1 + i] the join
// event of streams[i].
struct ForkRec { std::vector<hipStream_t> streams; std::vector<hipEvent_t> events; };
static std::map<std::pair<int, hipStream_t>, ForkRec> g_fork;

int fork_streams_for(hipStream_t main, int n, hipStream_t** streams, hipEvent_t** events) {
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_aux_mutex);
    ForkRec& r = g_fork[{dev, main}];
    while ((int)r.streams.size() < n) {
        hipStream_t st;
        SP1HIP_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        r.streams.push_back(st);
    }
    while ((int)r.events.size() < n + 1) {
        hipEvent_t e;
        SP1HIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        r.events.push_back(e);
    }
    *streams = r.streams.data();
    *events = r.events.data();
    return SP1HIP_SUCCESS;
}

// The caller's stream is about to be destroyed (a prover pool's slot; the caller has synchronised it): destroy the helper
// streams and events that were created for it and hand their cached arena blocks back. Without this every pool
// create / destroy cycle left one high-priority stream, its events and its arena blocks behind.
void release_stream_helpers(hipStream_t main) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    {
        std::lock_guard<std::mutex> lock(g_aux_mutex);
        auto a = g_aux.find({dev, main});
        if (a != g_aux.end()) {
            if (a->second.aux) streams.push_back(a->second.aux);
            events.insert(events.end(), a->second.events.begin(), a->second.events.end());
            g_aux.erase(a);
        }
        auto f = g_fork.find({dev, main});
        if (f != g_fork.end()) {
            streams.insert(streams.end(), f->second.streams.begin(), f->second.streams.end());
            events.insert(events.end(), f->second.events.begin(), f->second.events.end());
            g_fork.erase(f);
        }
    }
    for (hipStream_t st : streams) {
        (void)hipStreamSynchronize(st);
        (void)arena_release_stream(st);
        (void)hipStreamDestroy(st);
    }
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
}

// ---- stream-keyed buffer arena -------------------------------------------------------------------
// Steady-state proving allocates the same sizes over and over on one stream (codewords, trees, fold
// layers). Freed blocks are kept in a per-(device, stream, size) free list and handed back without
// touching the driver; reuse on the same stream is ordered by the stream itself. hipMallocAsync's own
// pool was measured to stall 80-120 ms per step on 0.5 GB blocks here (profiles/r01_notes.md).
struct ArenaKey {
    int device;
    hipStream_t stream;
    size_t bytes;
    bool operator<(const ArenaKey& o) const {
        if (device != o.device) return device < o.device;
        if (stream != o.stream) return stream < o.stream;
        return bytes < o.bytes;
    }
};
static std::mutex g_arena_mutex;
static std::map<ArenaKey, std::vector<void*>> g_arena;
static std::map<int, size_t> g_arena_cached_bytes;       // free-listed bytes per device (the cap is per device too)
// the size class a block was ALLOCATED with (>= 1 MiB blocks): a request may be served by a cached block of a slightly larger
// class (arena_alloc), and the block must return to its own list, not to the list of the request it last served
static std::unordered_map<void*, size_t> g_arena_block_class;
constexpr size_t ARENA_NEAR_FIT_MIN = (size_t)1 << 20;

// Size classes in steps of 1/8 of a power of two (<= 12.5 % internal slack): real shards have varying table heights,
// and exact-size keys would cache a new block for nearly every proof and never reuse it.
static size_t arena_round(size_t bytes) {
    if (bytes <= 4096) return bytes < 256 ? 256 : ((bytes + 255) / 256) * 256;
    int lg = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (lg - 3);
    return ((bytes + step - 1) / step) * step;
}
// cached (free-listed) bytes above this are handed back to the driver, largest blocks first: SP1HIP_ARENA_CAP_GB, default
// three quarters of the device's memory (216 GB on an MI355X: three provers in flight cache ~100 GB next to the caller's own stream). A core-shaped shard proof cycles ~34 GB of scratch per stream; with
// two provers in flight plus the blocks an earlier stream left behind, a 64 GB cap was crossed at the end of every proof
// and each crossing costs hipFree (a device synchronise) now and hipMalloc on the next proof — 234 ms per proof instead
// of ~85 whenever both provers' blocks met in the free lists.
static size_t arena_cap_bytes(int dev) {
    // computed once per device, with that device current (arena_alloc / arena_free run with the caller's device set)
    static std::map<int, size_t> caps;            // guarded by g_arena_mutex (every caller holds it)
    auto it = caps.find(dev);
    if (it != caps.end()) return it->second;
    size_t cap;
    if (const char* e = getenv("SP1HIP_ARENA_CAP_GB")) {
        // fractional values are legal ("0.5"); below 1 GiB every free would evict (hipFree = a device synchronise)
        const double gb = atof(e);
        cap = (size_t)(std::max(gb, 1.0) * (double)((size_t)1 << 30));
    } else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); cap = (size_t)64 << 30; }
        else cap = total_b / 4 * 3;
    }
    caps[dev] = cap;
    return cap;
}

size_t arena_trim_device(int dev);
// requests no cached block served (each one is a hipMalloc: a driver call that may synchronise the device) — SP1HIP_SHARD_TIMING prints them
static std::atomic<uint64_t> g_arena_misses{0}, g_arena_miss_bytes{0};
void arena_miss_stats(uint64_t* n, uint64_t* bytes) { *n = g_arena_misses.load(); *bytes = g_arena_miss_bytes.load(); }
int arena_alloc(void** ptr, size_t bytes, hipStream_t stream) {
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    const size_t sz = arena_round(bytes);
    {
        std::lock_guard<std::mutex> lock(g_arena_mutex);
        auto it = g_arena.find(ArenaKey{dev, stream, sz});
        if (it != g_arena.end() && !it->second.empty()) {
            *ptr = it->second.back();
            it->second.pop_back();
            g_arena_cached_bytes[dev] -= sz;
            return SP1HIP_SUCCESS;
        }
        // near fit: the smallest cached block of THIS stream in the next size classes (up to 1.5 x the request). The shards of
        // a real program all differ in their table heights (round 5, the 26 shards of the rsp block: 107 GB of cached blocks
        // after one pass on one stream, and with three provers in flight the device ran out of memory — a full trim, 1.6 s
        // of stall — every ten proofs); a request that misses its own class by one or two steps takes the neighbour's block
        if (sz >= ARENA_NEAR_FIT_MIN)
            for (auto nf = g_arena.upper_bound(ArenaKey{dev, stream, sz}); nf != g_arena.end() && nf->first.device == dev && nf->first.stream == stream && nf->first.bytes <= sz + sz / 2; ++nf)
                if (!nf->second.empty()) {
                    *ptr = nf->second.back();
                    nf->second.pop_back();
                    g_arena_cached_bytes[dev] -= nf->first.bytes;
                    return SP1HIP_SUCCESS;
                }
    }
    g_arena_misses.fetch_add(1, std::memory_order_relaxed);
    g_arena_miss_bytes.fetch_add(sz, std::memory_order_relaxed);
    hipError_t e = hipMalloc(ptr, sz);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        // 1. a block of this size class cached for ANOTHER stream of this device (free lists are per stream because reuse is
        //    ordered by the stream; a block changes streams only behind a synchronisation of the stream that last used it)
        hipStream_t donor = nullptr;
        void* stolen = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_arena_mutex);
            for (auto it = g_arena.begin(); it != g_arena.end(); ++it)
                if (it->first.device == dev && it->first.bytes == sz && it->first.stream != stream && !it->second.empty()) {
                    stolen = it->second.back();
                    it->second.pop_back();
                    g_arena_cached_bytes[dev] -= sz;
                    donor = it->first.stream;
                    break;
                }
        }
        if (stolen) {
            // (the NULL stream is a legal donor; a destroyed stream's blocks were released with it — arena_release_stream —
            // so a donor found in the free lists is normally alive)
            const hipError_t se = hipStreamSynchronize(donor);
            if (se == hipSuccess) { *ptr = stolen; return SP1HIP_SUCCESS; }
            (void)hipGetLastError();
            if (se == hipErrorInvalidHandle || se == hipErrorContextIsDestroyed || se == hipErrorInvalidResourceHandle) {
                *ptr = stolen;                // the donor stream is gone: nothing of it can still be running
                return SP1HIP_SUCCESS;
            }
            // any other failure (a sticky device fault, a lost device): the block may still be in use — it goes back to its
            // list and the error is the caller's
            {
                std::lock_guard<std::mutex> lock(g_arena_mutex);
                g_arena[ArenaKey{dev, donor, sz}].push_back(stolen);
                g_arena_cached_bytes[dev] += sz;
            }
            return map_hip_error(se, "hipStreamSynchronize(donor stream of a cached block)");
        }
        // 2. give THIS device's cached blocks back to the driver and retry once (other devices' provers are not stalled)
        arena_trim_device(dev);
        e = hipMalloc(ptr, sz);
    }
    SP1HIP_HIP(e);
    if (sz >= ARENA_NEAR_FIT_MIN) {
        std::lock_guard<std::mutex> lock(g_arena_mutex);
        g_arena_block_class[*ptr] = sz;
    }
    return SP1HIP_SUCCESS;
}

void arena_free(void* ptr, size_t bytes, hipStream_t stream) {
    if (!ptr) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    size_t sz = arena_round(bytes);
    std::vector<void*> evict;
    {
        std::lock_guard<std::mutex> lock(g_arena_mutex);
        if (sz >= ARENA_NEAR_FIT_MIN) {            // the class the block was allocated with (it may have served a smaller request)
            auto bc = g_arena_block_class.find(ptr);
            if (bc != g_arena_block_class.end()) sz = bc->second;
        }
        g_arena[ArenaKey{dev, stream, sz}].push_back(ptr);
        g_arena_cached_bytes[dev] += sz;
        // over the cap: drop this device's largest cached blocks (never the one just returned: it may still be in use by
        // work queued on its stream; the others were free-listed earlier, and hipFree waits for the device anyway)
        while (g_arena_cached_bytes[dev] > arena_cap_bytes(dev)) {
            auto best = g_arena.end();
            for (auto it = g_arena.begin(); it != g_arena.end(); ++it)
                if (it->first.device == dev && !it->second.empty() && !(it->second.size() == 1 && it->second.back() == ptr) &&
                    (best == g_arena.end() || it->first.bytes > best->first.bytes))
                    best = it;
            if (best == g_arena.end()) break;
            void* victim = best->second.front() == ptr ? best->second.back() : best->second.front();
            best->second.erase(std::find(best->second.begin(), best->second.end(), victim));
            g_arena_cached_bytes[dev] -= best->first.bytes;
            g_arena_block_class.erase(victim);
            evict.push_back(victim);
        }
    }
    for (void* p : evict) (void)hipFree(p);
}

// A stream that is about to be destroyed (a prover pool's slot): its cached blocks can never be reused — hand them back.
// The caller has synchronised the stream.
size_t arena_release_stream(hipStream_t stream) {
    std::vector<std::pair<void*, size_t>> blocks;
    {
        std::lock_guard<std::mutex> lock(g_arena_mutex);
        for (auto it = g_arena.begin(); it != g_arena.end();) {
            if (it->first.stream == stream) {
                for (void* p : it->second) { blocks.emplace_back(p, it->first.bytes); g_arena_cached_bytes[it->first.device] -= it->first.bytes; g_arena_block_class.erase(p); }
                it = g_arena.erase(it);
            } else {
                ++it;
            }
        }
    }
    size_t freed = 0;
    for (auto& b : blocks) { (void)hipFree(b.first); freed += b.second; }
    return freed;
}

// every cached block of one device (the current one must be `dev`: hipDeviceSynchronize / hipFree act on it)
size_t arena_trim_device(int dev) {
    std::vector<void*> blocks;
    size_t freed = 0;
    {
        std::lock_guard<std::mutex> lock(g_arena_mutex);
        for (auto it = g_arena.begin(); it != g_arena.end();) {
            if (it->first.device == dev) {
                for (void* p : it->second) { blocks.push_back(p); freed += it->first.bytes; g_arena_block_class.erase(p); }
                it = g_arena.erase(it);
            } else ++it;
        }
        g_arena_cached_bytes[dev] = 0;
    }
    (void)hipDeviceSynchronize();
    for (void* p : blocks) (void)hipFree(p);
    return freed;
}

// sp1hip_mem_trim: every device's cache
size_t arena_trim() {
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lock(g_arena_mutex);
        for (auto& kv : g_arena) if (std::find(devs.begin(), devs.end(), kv.first.device) == devs.end()) devs.push_back(kv.first.device);
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return 0;
    size_t freed = 0;
    for (int d : devs) {
        if (hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); continue; }
        freed += arena_trim_device(d);
    }
    (void)hipSetDevice(prev);
    return freed;
}

static std::mutex g_ctx_mutex;
static std::vector<DeviceCtx*> g_ctx;

int get_device_ctx(const DeviceCtx** out) {
    int dev = -1;
    SP1HIP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if ((int)g_ctx.size() <= dev) g_ctx.resize(dev + 1, nullptr);
    if (!g_ctx[dev]) {
        DeviceCtx* c = new DeviceCtx();
        c->device = dev;
        hipDeviceProp_t prop;
        SP1HIP_HIP(hipGetDeviceProperties(&prop, dev));
        c->num_cus = prop.multiProcessorCount;
        // keep freed stream-ordered allocations cached in the pool (hipMallocAsync/hipFreeAsync reuse)
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        p2::RoundConstants rc = p2::make_round_constants();
        SP1HIP_HIP(hipMalloc((void**)&c->d_rc, sizeof rc));
        SP1HIP_HIP(hipMemcpy(c->d_rc, &rc, sizeof rc, hipMemcpyHostToDevice));
        std::vector<uint32_t> lo(TW_LO), hi(TW_HI);
        uint32_t g = kb::two_adic_generator(kb::TWO_ADICITY), cur = kb::R1;
        for (int i = 0; i < TW_LO; i++) { lo[i] = cur; cur = kb::mul(cur, g); }
        uint32_t gh = cur;  // g^4096
        cur = kb::R1;
        for (int i = 0; i < TW_HI; i++) { hi[i] = cur; cur = kb::mul(cur, gh); }
        SP1HIP_HIP(hipMalloc((void**)&c->d_tw_lo, TW_LO * 4));
        SP1HIP_HIP(hipMalloc((void**)&c->d_tw_hi, TW_HI * 4));
        SP1HIP_HIP(hipMemcpy(c->d_tw_lo, lo.data(), TW_LO * 4, hipMemcpyHostToDevice));
        SP1HIP_HIP(hipMemcpy(c->d_tw_hi, hi.data(), TW_HI * 4, hipMemcpyHostToDevice));
        SP1HIP_HIP(hipDeviceSynchronize());      // (once per device: the tables are read from non-blocking streams)
        g_ctx[dev] = c;
    }
    *out = g_ctx[dev];
    return SP1HIP_SUCCESS;
}

// ---- round-sync slots (round_sync.hpp): (device counter, mapped pinned host words) pairs, recycled
static std::mutex g_rs_mutex;
static std::vector<RoundSyncSlot> g_rs_free[64];

int round_sync_acquire(RoundSyncSlot* out, hipStream_t stream) {
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    SP1HIP_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(g_rs_mutex);
        if (!g_rs_free[dev].empty()) { *out = g_rs_free[dev].back(); g_rs_free[dev].pop_back(); return SP1HIP_SUCCESS; }
    }
    RoundSyncSlot slot{nullptr, nullptr};
    SP1HIP_HIP(hipMalloc((void**)&slot.d_counter, RS_COUNTER_BYTES));
    // The counters must be zero before the first ticket. hipMemset is a fill kernel on the NULL stream and may return before it
    // has run; the provers' streams are non-blocking ones, which the NULL stream does not order — the first round kernel to take
    // tickets from a NEW slot could race the fill (seen once in ~1,000 pool proofs, round 4, fixed there with a device
    // synchronise: a stall on every prover's in-flight work). The fill now goes on the ACQUIRING prover's own stream: every kernel
    // that takes tickets from this slot is launched on that stream, or on a fork stream behind an event of it, after this call —
    // stream order is all it needs, nothing waits. (A private fill stream was tried first and cost 2 ms per proof: one more HIP
    // stream shifts the round-robin mapping of the prover's four streams onto the process's hardware queues — round 5)
    hipError_t e = hipMemsetAsync(slot.d_counter, 0, RS_COUNTER_BYTES, stream);
    if (e == hipSuccess) e = hipHostMalloc((void**)&slot.h_slot, RS_SLOT_WORDS * 4, hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(slot.d_counter);
        return map_hip_error(e, "creating a round-sync slot");
    }
    memset(slot.h_slot, 0, RS_SLOT_WORDS * 4);
    *out = slot;
    return SP1HIP_SUCCESS;
}

void round_sync_release(RoundSyncSlot slot) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lock(g_rs_mutex);
    g_rs_free[dev].push_back(slot);
}

__global__ __launch_bounds__(256) void mailbox_publish_kernel(const uint32_t* __restrict__ src, uint32_t n,
                                                              volatile uint32_t* slot, uint32_t seq) {
    for (uint32_t i = threadIdx.x; i < n; i += 256) slot[1 + i] = src[i];
    // the slot is uncached host memory: acknowledged stores are ordered before the sequence number; a system-scope fence
    // would also write back this XCD's whole L2 (round_sync.hpp)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) slot[0] = seq;
}

int mailbox_publish(const uint32_t* d_src, uint32_t n_words, uint32_t* h_slot, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(mailbox_publish_kernel, dim3(1), dim3(256), 0, s, d_src, n_words, (volatile uint32_t*)h_slot, seq);
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

static std::vector<MailboxSlot> g_mb_free[64];

int mailbox_acquire(MailboxSlot* out) {
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    SP1HIP_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(g_rs_mutex);
        if (!g_mb_free[dev].empty()) { *out = g_mb_free[dev].back(); g_mb_free[dev].pop_back(); return SP1HIP_SUCCESS; }
    }
    MailboxSlot slot{nullptr};
    SP1HIP_HIP(hipHostMalloc((void**)&slot.h_slot, (size_t)(MAILBOX_WORDS + 1) * 4, hipHostMallocMapped));
    memset(slot.h_slot, 0, (size_t)(MAILBOX_WORDS + 1) * 4);
    *out = slot;
    return SP1HIP_SUCCESS;
}

void mailbox_release(MailboxSlot slot) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lock(g_rs_mutex);
    g_mb_free[dev].push_back(slot);
}

static std::vector<HostGateBlock> g_gate_free[64];

int host_gate_acquire(HostGateBlock* out) {
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    SP1HIP_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(g_rs_mutex);
        if (!g_gate_free[dev].empty()) { *out = g_gate_free[dev].back(); g_gate_free[dev].pop_back(); return SP1HIP_SUCCESS; }
    }
    HostGateBlock b{nullptr};
    const size_t words = 64 + (size_t)GATE_RING * GATE_SLOT_WORDS;
    SP1HIP_HIP(hipHostMalloc((void**)&b.h, words * 4, hipHostMallocMapped));
    memset(b.h, 0, words * 4);
    *out = b;
    return SP1HIP_SUCCESS;
}

void host_gate_release(HostGateBlock b) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lock(g_rs_mutex);
    g_gate_free[dev].push_back(b);
}

static std::vector<PinnedBlock> g_pin_free[64];

int pinned_stage_acquire(PinnedBlock* out) {
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    SP1HIP_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(g_rs_mutex);
        if (!g_pin_free[dev].empty()) { *out = g_pin_free[dev].back(); g_pin_free[dev].pop_back(); return SP1HIP_SUCCESS; }
    }
    PinnedBlock b{nullptr};
    SP1HIP_HIP(hipHostMalloc((void**)&b.h, PINNED_STAGE_BYTES, hipHostMallocDefault));
    *out = b;
    return SP1HIP_SUCCESS;
}

void pinned_stage_release(PinnedBlock b) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lock(g_rs_mutex);
    g_pin_free[dev].push_back(b);
}

}  // namespace sp1hip

using namespace sp1hip;

namespace sp1hip {
namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        const char* e = getenv("SP1HIP_ROCTX");
        if (e && e[0] == '0') return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const RoctxApi& roctx_api() { static const RoctxApi api; return api; }
}  // namespace
void roctx_push(const char* name) { const RoctxApi& a = roctx_api(); if (a.push) a.push(name); }
void roctx_pop() { const RoctxApi& a = roctx_api(); if (a.pop) a.pop(); }
}  // namespace sp1hip

extern "C" {

const char* sp1hip_last_error(void) { return g_last_error.c_str(); }
const char* sp1hip_version(void) { return "sp1hip 0.1.0 (gfx950; KoalaBear; Poseidon2-16)"; }

int sp1hip_timers_enable(int on) { g_timers_on = on != 0; return SP1HIP_SUCCESS; }
int sp1hip_timers_reset(void) {
    std::lock_guard<std::mutex> lock(g_timer_mutex);
    for (auto& r : g_timer_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_timer_recs.clear();
    return SP1HIP_SUCCESS;
}
int sp1hip_timers_read(const char* name, uint64_t* launches, double* total_ms) {
    SP1HIP_REQUIRE(name && launches && total_ms, "null argument");
    std::lock_guard<std::mutex> lock(g_timer_mutex);
    *launches = 0;
    *total_ms = 0;
    for (auto& r : g_timer_recs) {
        if (r.name != name) continue;
        SP1HIP_HIP(hipEventSynchronize(r.b));
        float ms = 0;
        SP1HIP_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        *launches += 1;
        *total_ms += ms;
    }
    return SP1HIP_SUCCESS;
}

int sp1hip_stream_release(sp1hip_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SP1HIP_REQUIRE(s != nullptr, "the default stream is never released");
    SP1HIP_HIP(hipStreamSynchronize(s));
    release_stream_helpers(s);
    (void)arena_release_stream(s);
    return SP1HIP_SUCCESS;
}

int sp1hip_mem_trim(size_t* released_bytes) {
    const size_t n = arena_trim();
    if (released_bytes) *released_bytes = n;
    return SP1HIP_SUCCESS;
}

int sp1hip_device_count(int* count) {
    SP1HIP_REQUIRE(count, "null count");
    hipError_t e = hipGetDeviceCount(count);
    if (e == hipErrorNoDevice) { *count = 0; (void)hipGetLastError(); return SP1HIP_SUCCESS; }
    SP1HIP_HIP(e);
    return SP1HIP_SUCCESS;
}
int sp1hip_set_device(int device) { SP1HIP_HIP(hipSetDevice(device)); return SP1HIP_SUCCESS; }
int sp1hip_get_device(int* device) { SP1HIP_REQUIRE(device, "null"); SP1HIP_HIP(hipGetDevice(device)); return SP1HIP_SUCCESS; }
int sp1hip_mem_info(size_t* free_bytes, size_t* total_bytes) {
    SP1HIP_REQUIRE(free_bytes && total_bytes, "null output");
    SP1HIP_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return SP1HIP_SUCCESS;
}
int sp1hip_malloc(void** d_ptr, size_t bytes) {
    SP1HIP_REQUIRE(d_ptr, "null output");
    SP1HIP_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return SP1HIP_SUCCESS;
}
int sp1hip_free(void* d_ptr) { SP1HIP_HIP(hipFree(d_ptr)); return SP1HIP_SUCCESS; }
int sp1hip_malloc_async(void** d_ptr, size_t bytes, sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(d_ptr, "null output");
    SP1HIP_HIP(hipMallocAsync(d_ptr, bytes ? bytes : 1, S(stream)));
    return SP1HIP_SUCCESS;
}
int sp1hip_free_async(void* d_ptr, sp1hip_stream_t stream) { SP1HIP_HIP(hipFreeAsync(d_ptr, S(stream))); return SP1HIP_SUCCESS; }
int sp1hip_malloc_host(void** h_ptr, size_t bytes) {
    SP1HIP_REQUIRE(h_ptr, "null output");
    SP1HIP_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return SP1HIP_SUCCESS;
}
int sp1hip_free_host(void* h_ptr) { SP1HIP_HIP(hipHostFree(h_ptr)); return SP1HIP_SUCCESS; }
int sp1hip_memcpy_h2d_async(void* d, const void* h, size_t n, sp1hip_stream_t s) {
    SP1HIP_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, S(s)));
    return SP1HIP_SUCCESS;
}
int sp1hip_memcpy_d2h_async(void* h, const void* d, size_t n, sp1hip_stream_t s) {
    SP1HIP_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, S(s)));
    return SP1HIP_SUCCESS;
}
int sp1hip_memcpy_d2d_async(void* d, const void* src, size_t n, sp1hip_stream_t s) {
    SP1HIP_HIP(hipMemcpyAsync(d, src, n, hipMemcpyDeviceToDevice, S(s)));
    return SP1HIP_SUCCESS;
}
int sp1hip_memset_async(void* d, int v, size_t n, sp1hip_stream_t s) {
    SP1HIP_HIP(hipMemsetAsync(d, v, n, S(s)));
    return SP1HIP_SUCCESS;
}
int sp1hip_stream_create(sp1hip_stream_t* stream) {
    SP1HIP_REQUIRE(stream, "null output");
    hipStream_t s;
    SP1HIP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return SP1HIP_SUCCESS;
}
int sp1hip_stream_destroy(sp1hip_stream_t stream) {
    if (stream) SP1HIP_TRY(sp1hip_stream_release(stream));     // its helper streams, events and cached buffers go with it
    SP1HIP_HIP(hipStreamDestroy(S(stream)));
    return SP1HIP_SUCCESS;
}
int sp1hip_stream_synchronize(sp1hip_stream_t stream) { SP1HIP_HIP(hipStreamSynchronize(S(stream))); return SP1HIP_SUCCESS; }
int sp1hip_stream_query(sp1hip_stream_t stream) {
    hipError_t e = hipStreamQuery(S(stream));
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return SP1HIP_ERROR_NOT_READY; }
    SP1HIP_HIP(e);
    return SP1HIP_SUCCESS;
}
int sp1hip_event_create(sp1hip_event_t* event) {
    SP1HIP_REQUIRE(event, "null output");
    hipEvent_t e;
    SP1HIP_HIP(hipEventCreate(&e));
    *event = e;
    return SP1HIP_SUCCESS;
}
int sp1hip_event_destroy(sp1hip_event_t event) { SP1HIP_HIP(hipEventDestroy((hipEvent_t)event)); return SP1HIP_SUCCESS; }
int sp1hip_event_record(sp1hip_event_t event, sp1hip_stream_t stream) {
    SP1HIP_HIP(hipEventRecord((hipEvent_t)event, S(stream)));
    return SP1HIP_SUCCESS;
}
int sp1hip_event_synchronize(sp1hip_event_t event) { SP1HIP_HIP(hipEventSynchronize((hipEvent_t)event)); return SP1HIP_SUCCESS; }
int sp1hip_event_elapsed_ms(float* ms, sp1hip_event_t start, sp1hip_event_t stop) {
    SP1HIP_REQUIRE(ms, "null output");
    SP1HIP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SP1HIP_SUCCESS;
}
int sp1hip_stream_wait_event(sp1hip_stream_t stream, sp1hip_event_t event) {
    SP1HIP_HIP(hipStreamWaitEvent(S(stream), (hipEvent_t)event, 0));
    return SP1HIP_SUCCESS;
}

}  // extern "C"
