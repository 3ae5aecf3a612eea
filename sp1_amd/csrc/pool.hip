// sp1_amd/csrc/pool.hip — a prover pool: N shard proofs in flight on one GPU, staging included.
//
// One shard proof leaves the GPU idle for 20-25 % of its duration (~120 transcript hand-overs per
// LogUp-GKR stage, the host rounds of every layer, BaseFold's chains of small dependent launches). The library is re-entrant
// per stream, so a second and a third proof fill those gaps — but until now "several in flight" existed only as Python
// threads in bench.py. The pool is that as an object of the C ABI:
//   * n_slots PROVER SLOTS, each a host thread + a HIP stream (+ the stream-keyed arena share, round-sync / mailbox / pinned
//     slots the provers pool per stream): a slot proves one shard at a time, `sp1hip_prove_shard_with_pk` on its stream;
//   * one STAGER (thread + stream): host traces of the next shards go up — `sp1hip_stage_tables`: PCIe copies overlapped with
//     the on-GPU transposes — while the slots prove; at most n_slots + 1 shards are staged ahead;
//   * tickets: submit returns at once, wait blocks for one proof.
// It is the counterpart of the reference's `ProverSemaphore` (/root/reference/crates/hypercube/src/prover/permits.rs:L36-L66;
// the GPU worker builder takes 1 permit, /root/reference/sp1-gpu/crates/prover_components/src/builder.rs:L107) together with
// the `trace_buffers` worker queue of its shard prover (sp1-gpu/crates/prover_components/src/components.rs:L100-L108), moved
// below the FFI so that every host language gets it.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

namespace sp1hip {
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Job {
    uint64_t ticket = 0;
    const sp1hip_pk_t* pk = nullptr;
    std::vector<sp1hip_pool_chip_t> chips;      // pointers inside stay caller-owned until the ticket has been waited for
    std::vector<uint32_t> publics;
    std::vector<void*> staged;                  // per chip: device table the stager filled (or nullptr)
    std::vector<size_t> staged_bytes;
    hipEvent_t staged_ev = nullptr;
    bool any_staged = false;
    // result
    bool done = false;
    int status = SP1HIP_ERROR_NOT_READY;
    std::string error;
    std::vector<uint8_t> proof;
    double t_submit = 0, t_staged = 0, t_start = 0, t_done = 0;
    int slot = -1;
};

}  // namespace
}  // namespace sp1hip

using namespace sp1hip;

struct sp1hip_pool_s {
    int device = 0, n_slots = 0;
    std::mutex m;
    std::condition_variable cv_in, cv_ready, cv_done, cv_room;
    std::deque<std::shared_ptr<Job>> in, ready;
    std::map<uint64_t, std::shared_ptr<Job>> jobs;
    uint64_t next_ticket = 1;
    bool stop = false;
    int staged_ahead = 0;                       // shards staged (or being staged) and not yet finished
    hipStream_t stage_stream = nullptr;
    std::vector<hipStream_t> slot_streams;
    std::thread stager;
    std::vector<std::thread> workers;

    void fail(const std::shared_ptr<Job>& j, int status) {
        j->error = sp1hip_last_error();
        j->status = status;
    }

    void free_staged(const std::shared_ptr<Job>& j) {
        for (size_t c = 0; c < j->staged.size(); c++)
            if (j->staged[c]) arena_free(j->staged[c], j->staged_bytes[c], stage_stream);
        j->staged.clear();
        if (j->staged_ev) { (void)hipEventDestroy(j->staged_ev); j->staged_ev = nullptr; }
    }

    void finish(const std::shared_ptr<Job>& j) {
        // (a finished prove call has handed its last bytes to the host: every kernel that read the staged tables is done)
        free_staged(j);
        j->t_done = now_ms();
        {
            std::lock_guard<std::mutex> lk(m);
            j->done = true;
            if (j->any_staged) staged_ahead--;
        }
        cv_done.notify_all();
        cv_room.notify_all();
    }

    // ---- the stager: submitted shards -> device tables, in submission order
    void stager_main() {
        (void)hipSetDevice(device);
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_in.wait(lk, [&] { return stop || !in.empty(); });
                if (in.empty()) return;          // stop
                j = in.front();
                in.pop_front();
                if (j->any_staged) {
                    cv_room.wait(lk, [&] { return stop || staged_ahead <= n_slots; });
                    if (stop) { lk.unlock(); j->status = SP1HIP_ERROR_RUNTIME; j->error = "pool destroyed"; finish_unstaged(j); continue; }
                    staged_ahead++;
                }
            }
            int st = SP1HIP_SUCCESS;
            if (j->any_staged) st = stage(j);
            j->t_staged = now_ms();
            if (st != SP1HIP_SUCCESS) { fail(j, st); finish(j); continue; }
            {
                std::lock_guard<std::mutex> lk(m);
                ready.push_back(j);
            }
            cv_ready.notify_one();
        }
    }
    void finish_unstaged(const std::shared_ptr<Job>& j) {
        j->any_staged = false;
        finish(j);
    }

    int stage(const std::shared_ptr<Job>& j) {
        const size_t n = j->chips.size();
        j->staged.assign(n, nullptr);
        j->staged_bytes.assign(n, 0);
        std::vector<sp1hip_host_table_t> host;
        std::vector<uint32_t*> dst;
        for (size_t c = 0; c < n; c++) {
            const sp1hip_pool_chip_t& ch = j->chips[c];
            if (!ch.h_main || ch.real_rows == 0 || ch.main_width == 0) continue;
            const size_t bytes = (size_t)ch.real_rows * ch.main_width * 4;
            SP1HIP_TRY(arena_alloc(&j->staged[c], bytes, stage_stream));
            j->staged_bytes[c] = bytes;
            host.push_back(sp1hip_host_table_t{ch.h_main, ch.real_rows, ch.main_width});
            dst.push_back((uint32_t*)j->staged[c]);
        }
        if (!host.empty()) SP1HIP_TRY(sp1hip_stage_tables(host.data(), (int)host.size(), dst.data(), stage_stream));
        SP1HIP_HIP(hipEventCreateWithFlags(&j->staged_ev, hipEventDisableTiming));
        SP1HIP_HIP(hipEventRecord(j->staged_ev, stage_stream));
        return SP1HIP_SUCCESS;
    }

    // ---- a prover slot
    void worker_main(int slot) {
        (void)hipSetDevice(device);
        hipStream_t s = slot_streams[slot];
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_ready.wait(lk, [&] { return stop || !ready.empty(); });
                if (ready.empty()) return;       // stop
                j = ready.front();
                ready.pop_front();
            }
            j->slot = slot;
            j->t_start = now_ms();
            const int st = prove(j, s);
            if (st != SP1HIP_SUCCESS) fail(j, st); else j->status = SP1HIP_SUCCESS;
            finish(j);
        }
    }

    int prove(const std::shared_ptr<Job>& j, hipStream_t s) {
        if (j->staged_ev) SP1HIP_HIP(hipStreamWaitEvent(s, j->staged_ev, 0));
        std::vector<sp1hip_shard_chip_t> chips(j->chips.size());
        for (size_t c = 0; c < chips.size(); c++) {
            const sp1hip_pool_chip_t& p = j->chips[c];
            const uint32_t* d_main = !j->staged.empty() && j->staged[c] ? (const uint32_t*)j->staged[c] : p.d_main;
            chips[c] = sp1hip_shard_chip_t{p.name, p.program, p.n_instr, p.num_constraints, p.interactions, p.n_words,
                                           p.main_width, p.prep_width, d_main, p.d_prep, p.real_rows};
        }
        const uint32_t* pub = j->publics.empty() ? nullptr : j->publics.data();
        size_t need = 0;
        const int q = sp1hip_prove_shard_with_pk(j->pk, chips.data(), (int)chips.size(), pub, (int)j->publics.size(), nullptr, 0,
                                                 nullptr, &need, s);
        if (q == SP1HIP_SUCCESS) {            // (the size query is specified to fail with BUFFER_TOO_SMALL)
            set_error("internal error: the proof size query succeeded without a buffer");
            return SP1HIP_ERROR_RUNTIME;
        }
        if (q != SP1HIP_ERROR_BUFFER_TOO_SMALL) return q;
        j->proof.resize(need);
        size_t len = need;
        SP1HIP_TRY(sp1hip_prove_shard_with_pk(j->pk, chips.data(), (int)chips.size(), pub, (int)j->publics.size(), nullptr, 0,
                                              j->proof.data(), &len, s));
        j->proof.resize(len);
        return SP1HIP_SUCCESS;
    }
};

extern "C" {

int sp1hip_pool_create(int device, int n_slots, sp1hip_pool_t** out) {
    SP1HIP_REQUIRE(out && n_slots >= 1 && n_slots <= 16, "n_slots must be 1 .. 16");
    int count = 0;
    SP1HIP_HIP(hipGetDeviceCount(&count));
    SP1HIP_REQUIRE(device >= 0 && device < count, "no such device");
    int prev = 0;
    SP1HIP_HIP(hipGetDevice(&prev));
    SP1HIP_HIP(hipSetDevice(device));
    std::unique_ptr<sp1hip_pool_s> p(new sp1hip_pool_s());
    p->device = device;
    p->n_slots = n_slots;
    // Nothing may leak or cross the C ABI as an exception when creation fails half-way: streams made so far are destroyed,
    // threads started so far are stopped and joined, and the failure comes back as a status code.
    sp1hip_pool_s* raw = p.get();
    auto undo = [&](int status) {
        {
            std::lock_guard<std::mutex> lk(raw->m);
            raw->stop = true;
        }
        raw->cv_in.notify_all();
        raw->cv_ready.notify_all();
        raw->cv_room.notify_all();
        if (raw->stager.joinable()) raw->stager.join();
        for (auto& w : raw->workers) if (w.joinable()) w.join();
        if (raw->stage_stream) (void)hipStreamDestroy(raw->stage_stream);
        for (hipStream_t st : raw->slot_streams) if (st) (void)hipStreamDestroy(st);
        (void)hipSetDevice(prev);
        return status;
    };
    hipError_t he = hipStreamCreateWithFlags(&p->stage_stream, hipStreamNonBlocking);
    p->slot_streams.assign(n_slots, nullptr);
    for (int i = 0; i < n_slots && he == hipSuccess; i++) he = hipStreamCreateWithFlags(&p->slot_streams[i], hipStreamNonBlocking);
    if (he != hipSuccess) return undo(map_hip_error(he, "sp1hip_pool_create: stream creation failed"));
    try {
        p->stager = std::thread([raw] { raw->stager_main(); });
        for (int i = 0; i < n_slots; i++) p->workers.emplace_back([raw, i] { raw->worker_main(i); });
    } catch (const std::exception& e) {
        set_error("sp1hip_pool_create: could not start the pool's threads (%s)", e.what());
        return undo(SP1HIP_ERROR_RUNTIME);
    }
    (void)hipSetDevice(prev);
    *out = p.release();
    return SP1HIP_SUCCESS;
}

void sp1hip_pool_destroy(sp1hip_pool_t* pool) {
    if (!pool) return;
    {
        // outstanding work is finished first (submitted shards are proven; their tickets simply go unclaimed)
        std::unique_lock<std::mutex> lk(pool->m);
        pool->cv_done.wait(lk, [&] {
            for (auto& kv : pool->jobs) if (!kv.second->done) return false;
            return true;
        });
        pool->stop = true;
    }
    pool->cv_in.notify_all();
    pool->cv_ready.notify_all();
    pool->cv_room.notify_all();
    if (pool->stager.joinable()) pool->stager.join();
    for (auto& w : pool->workers) if (w.joinable()) w.join();
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(pool->device);
    // the slots' scratch (a core shard proof cycles ~34 GB per stream) is keyed by stream in the arena: give it back, a
    // destroyed stream can never reuse it
    // (and so are the helper streams created for them: the commit's encode stream, the GKR's descriptor stream, the
    // zerocheck's fork streams — each with events and arena blocks of its own)
    (void)hipStreamSynchronize(pool->stage_stream);
    release_stream_helpers(pool->stage_stream);
    (void)arena_release_stream(pool->stage_stream);
    (void)hipStreamDestroy(pool->stage_stream);
    for (hipStream_t s : pool->slot_streams) {
        (void)hipStreamSynchronize(s);
        release_stream_helpers(s);
        (void)arena_release_stream(s);
        (void)hipStreamDestroy(s);
    }
    (void)hipSetDevice(prev);
    delete pool;
}

int sp1hip_pool_submit(sp1hip_pool_t* pool, const sp1hip_pk_t* pk, const sp1hip_pool_chip_t* chips, int n_chips,
                       const uint32_t* h_publics, int n_publics, sp1hip_ticket_t* ticket) {
    SP1HIP_REQUIRE(pool && pk && chips && n_chips > 0 && ticket && n_publics >= 0 && (n_publics == 0 || h_publics), "bad argument");
    auto j = std::make_shared<Job>();
    j->pk = pk;
    j->chips.assign(chips, chips + n_chips);
    if (n_publics) j->publics.assign(h_publics, h_publics + n_publics);
    for (const sp1hip_pool_chip_t& c : j->chips) {
        SP1HIP_REQUIRE(c.name && c.interactions, "null chip field");
        SP1HIP_REQUIRE(!(c.h_main && c.d_main), "a chip's main trace is given either on the host or on the device");
        if (c.h_main && c.real_rows && c.main_width) j->any_staged = true;
    }
    j->t_submit = now_ms();
    {
        std::lock_guard<std::mutex> lk(pool->m);
        SP1HIP_REQUIRE(!pool->stop, "pool is shutting down");
        j->ticket = pool->next_ticket++;
        pool->jobs[j->ticket] = j;
        pool->in.push_back(j);
    }
    pool->cv_in.notify_one();
    *ticket = j->ticket;
    return SP1HIP_SUCCESS;
}

static int pool_collect(sp1hip_pool_t* pool, sp1hip_ticket_t ticket, bool block, uint8_t* h_proof, size_t* proof_len,
                        sp1hip_pool_times_t* times) {
    SP1HIP_REQUIRE(pool && proof_len, "bad argument");
    std::shared_ptr<Job> j;
    {
        std::unique_lock<std::mutex> lk(pool->m);
        auto it = pool->jobs.find(ticket);
        SP1HIP_REQUIRE(it != pool->jobs.end(), "unknown (or already collected) ticket");
        j = it->second;
        if (!j->done) {
            if (!block) { set_error("sp1hip_pool_try_wait: ticket %llu is still in flight", (unsigned long long)ticket); return SP1HIP_ERROR_NOT_READY; }
            pool->cv_done.wait(lk, [&] { return j->done; });
        }
        if (j->status == SP1HIP_SUCCESS && (!h_proof || *proof_len < j->proof.size())) {
            *proof_len = j->proof.size();       // the ticket stays claimable
            set_error("sp1hip_pool_wait: proof buffer too small, need %zu bytes", j->proof.size());
            return SP1HIP_ERROR_BUFFER_TOO_SMALL;
        }
        pool->jobs.erase(it);
    }
    if (times) *times = sp1hip_pool_times_t{j->t_staged - j->t_submit, j->t_start - j->t_staged, j->t_done - j->t_start, j->slot};
    if (j->status != SP1HIP_SUCCESS) {
        set_error("pool ticket %llu: %s", (unsigned long long)ticket, j->error.c_str());
        return j->status;
    }
    memcpy(h_proof, j->proof.data(), j->proof.size());
    *proof_len = j->proof.size();
    return SP1HIP_SUCCESS;
}

int sp1hip_pool_wait(sp1hip_pool_t* pool, sp1hip_ticket_t ticket, uint8_t* h_proof, size_t* proof_len, sp1hip_pool_times_t* times) {
    return pool_collect(pool, ticket, true, h_proof, proof_len, times);
}

int sp1hip_pool_try_wait(sp1hip_pool_t* pool, sp1hip_ticket_t ticket, uint8_t* h_proof, size_t* proof_len, sp1hip_pool_times_t* times) {
    return pool_collect(pool, ticket, false, h_proof, proof_len, times);
}

}  // extern "C"
