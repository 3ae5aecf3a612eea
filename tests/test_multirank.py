"""world_size-2 gloo test (CPU) of the N > 1 path: shard striping, max-over-ranks timing and the proof
blob exchange. Per-shard "proving" is replaced by the host-side transcript of the product library
(no kernels needed), so the distributed plumbing is exercised exactly as bench.py / a multi-GPU
controller would use it."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _shard_blob(idx):
    """Deterministic stand-in for a shard proof: the transcript digest of the shard index."""
    from sp1_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.sp1hip_challenger_new(C.byref(h)) == 0
    xs = np.arange(idx, idx + 5, dtype=np.uint32)
    assert lib.sp1hip_challenger_observe(h, xs.ctypes.data_as(C.POINTER(C.c_uint32)), xs.size) == 0
    out = []
    for _ in range(3 + idx % 4):                     # variable-length blobs
        v = C.c_uint32()
        assert lib.sp1hip_challenger_sample(h, C.byref(v)) == 0
        out.append(v.value)
    lib.sp1hip_challenger_free(h)
    return np.array(out, dtype=np.uint32).tobytes()


def _worker(rank, world, port, n_shards, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sp1_amd import shards
    mine = shards.stripe(n_shards, world, rank)
    blobs = {i: _shard_blob(i) for i in mine}
    merged = shards.gather_blobs(blobs)
    t = shards.max_over_ranks(1.0 + rank)
    dist.barrier()
    q.put((rank, mine, sorted(merged.items()), t))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_shards", [(2, 7), (2, 1), (2, 0), (8, 19), (8, 5)])
def test_striping_and_blob_exchange_gloo(world, n_shards):
    """world 8 = BASELINE's node (configs 4 and 5): striping with more ranks than shards (8 ranks, 5 shards) included."""
    import __graft_entry__ as g
    g.build_hip()
    port = 29500 + (os.getpid() % 2000) + n_shards + 37 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_shards, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sorted((i, _shard_blob(i)) for i in range(n_shards))
    assigned = sorted(i for _, mine, _, _ in results for i in mine)
    assert assigned == list(range(n_shards))                      # every shard proven exactly once
    for rank, mine, merged, t in results:
        assert mine == list(range(rank, n_shards, world))
        assert merged == want                                      # every rank ends with every proof blob
        assert t == float(world)                                   # max over ranks of 1 + rank


def _combine(kids):
    """Stand-in for proving a parent node: an order-sensitive digest of the children's bytes."""
    import hashlib
    return hashlib.sha256(b"|".join(kids)).digest() + bytes([len(kids)])


def _tree_worker(rank, world, port, n_shards, arity, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sp1_amd import shards
    blobs = {i: _shard_blob(i) for i in shards.stripe(n_shards, world, rank)}
    root = shards.reduce_tree(blobs, n_shards, _combine, arity)
    dist.barrier()
    q.put((rank, root))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_shards,arity", [(2, 7, 2), (2, 8, 3), (3, 5, 2), (2, 1, 2), (8, 21, 2), (8, 8, 3), (8, 3, 2)])
def test_compress_tree_over_ranks_gloo(world, n_shards, arity):
    """The recursion-tree reduce step: point-to-point proof movement to the parent's rank, same root as one process."""
    import __graft_entry__ as g
    g.build_hip()
    from sp1_amd import shards
    want = shards.reduce_tree({i: _shard_blob(i) for i in range(n_shards)}, n_shards, _combine, arity)
    port = 31500 + (os.getpid() % 2000) + 10 * n_shards + arity + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tree_worker, args=(r, world, port, n_shards, arity, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] == want and want is not None
    assert all(results[r] is None for r in range(1, world))


def test_single_process_helpers():
    from sp1_amd import shards
    assert shards.stripe(5, 1, 0) == [0, 1, 2, 3, 4]
    assert shards.max_over_ranks(3.5) == 3.5
    assert shards.gather_blobs({2: b"ab"}) == {2: b"ab"}


# ---- the compress tree with REAL proofs: every node is a ShardProof of the reference's recursion machine ------------------
_COUNTS = {"BaseAlu": 20, "ExtAlu": 20, "MemoryConst": 30, "MemoryVar": 10, "Poseidon2WideDeg3": 4, "PrefixSumChecks": 8, "Select": 20}
_L, _LSH, _BATCH, _FRI = 7, 6, 4, (1, 5, 4)


def _oracle_prove(tables, publics):
    """prove(tables, publics) for shards.recursion_combine on a rank without a GPU: the CPU oracle (tests may use it). The blob
    that travels is the node's preprocessed commitment (its verifying key) followed by bincode(ShardProof)."""
    import pyoracle as orc
    from sp1_amd.machines import recursion as R
    chips = [(a, i, tables[a.name][1], tables[a.name][0]) for a, i in R.compress_machine()]
    prep = orc.JaggedRound([c[3] for c in chips], _L, _LSH, _BATCH, _FRI[0])
    ch = orc.Challenger()
    ch.observe(prep.commit)
    orc.set_gkr_sparse(True)                         # the jagged-aware oracle prover: the same bytes as the dense one, 7x faster here
    try:
        return prep.commit.tobytes() + orc.shard_prove(chips, publics, prep, _L, _LSH, _BATCH, ch, *_FRI)
    finally:
        orc.set_gkr_sparse(False)


def _verify_node(blob):
    import numpy as np
    import pyoracle as orc
    from sp1_amd.machines import recursion as R
    commit = np.frombuffer(blob[:32], dtype=np.uint32).copy()
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32)) for a, i in R.compress_machine()]
    ch = orc.Challenger()
    ch.observe(commit)
    return orc.shard_verify(shapes, commit, blob[32:], _L, _LSH, ch, *_FRI)


def _leaf_proof(i):
    from sp1_amd.machines import recursion_trace as RT
    tables, publics = RT.generate(_COUNTS, seed=100 + i)
    return _oracle_prove(tables, publics)


def _real_tree_worker(rank, world, port, n_leaves, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle as orc
    orc.set_threads(2)
    from sp1_amd import shards
    leaves = {i: _leaf_proof(i) for i in shards.stripe(n_leaves, world, rank)}
    root = shards.reduce_tree(leaves, n_leaves, shards.recursion_combine(_oracle_prove, _COUNTS, seed=7), 2)
    dist.barrier()
    q.put((rank, root))
    dist.destroy_process_group()


def test_compress_tree_moves_and_verifies_real_recursion_proofs_gloo():
    """world 2, 3 leaves: every leaf and every parent is a real ShardProof over the reference's recursion compress machine
    (the transcription the reference's own proof pins); parents commit to their children through the public-values digest
    (`shards.recursion_combine`). The root that arrives on rank 0 is byte-identical to the single-process tree, the full
    verifier accepts it, and its committed digest is the hash of the two blobs that were sent to its rank."""
    import hashlib
    import struct

    import __graft_entry__ as g
    g.build_hip()
    g.build_oracle()
    from sp1_amd import shards
    from sp1_amd.machines import recursion as R
    n_leaves, world = 3, 2
    combine = shards.recursion_combine(_oracle_prove, _COUNTS, seed=7)
    leaves = {i: _leaf_proof(i) for i in range(n_leaves)}
    want = shards.reduce_tree(leaves, n_leaves, combine, 2)
    assert all(_verify_node(b) == 0 for b in leaves.values()) and _verify_node(want) == 0
    # the root's committed digest = sha256 of its children [parent(leaf0, leaf1), leaf2]
    kids = [combine([leaves[0], leaves[1]]), leaves[2]]
    h = hashlib.sha256()
    for c in kids:
        h.update(len(c).to_bytes(8, "little"))
        h.update(c)
    d = h.digest()
    digest = [int.from_bytes(d[4 * i:4 * i + 4], "little") % 0x7F000001 for i in range(8)]
    proof = want[32:]
    assert struct.unpack_from("<Q", proof, 0)[0] == R.NUM_PUBLIC_VALUES
    assert list(struct.unpack_from("<8I", proof, 8 + 4 * R.PV_DIGEST_OFFSET)) == digest
    port = 33500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_real_tree_worker, args=(r, world, port, n_leaves, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] == want and results[1] is None
    bad = bytearray(want)
    bad[32 + 8 + 4 * R.PV_DIGEST_OFFSET] ^= 1                      # a parent that lies about its children is rejected
    assert _verify_node(bytes(bad)) != 0


def _real_tree_worker_levels(rank, world, port, n_leaves, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1", OMP_WAIT_POLICY="passive")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle as orc
    orc.set_threads(1)                                # eight oracle processes on one box
    from sp1_amd import shards
    levels = []
    leaves = {i: _leaf_proof(i) for i in shards.stripe(n_leaves, world, rank)}
    root = shards.reduce_tree(leaves, n_leaves, shards.recursion_combine(_oracle_prove, _COUNTS, seed=7), 2,
                              on_level=lambda li, st: levels.append((li, st["nodes"], st["parents"], st["proved"], st["sent_bytes"], st["recv_bytes"])))
    dist.barrier()
    q.put((rank, root, levels))
    dist.destroy_process_group()


def test_compress_tree_at_world_8_fully_verifies_the_root_gloo():
    """world 8, 9 leaves (more leaves than ranks, ragged levels 9 -> 5 -> 3 -> 2 -> 1): leaves striped over eight
    processes, children sent point to point to their parent's rank, every node a real ShardProof of the recursion compress machine.
    The oracle's FULL verifier (constraints + interactions of the transcribed machine) accepts the root that arrives on rank 0, the
    root commits to its two children (its public-values digest is their hash: the world-2 test checks that chain byte for byte
    against the single-process tree), and the per-level callback accounts for every node and every byte that moved."""
    import __graft_entry__ as g
    g.build_hip()
    g.build_oracle()
    n_leaves, world = 9, 8
    port = 35500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_real_tree_worker_levels, args=(r, world, port, n_leaves, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r: (root, lv) for r, root, lv in (q.get(timeout=900) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    root = results[0][0]
    assert root is not None and all(results[r][0] is None for r in range(1, world))
    assert _verify_node(root) == 0
    shape = [(lv[0], lv[1], lv[2]) for lv in results[0][1]]
    assert shape == [(0, 9, 5), (1, 5, 3), (2, 3, 2), (3, 2, 1)]
    for li, want_proved in enumerate([4, 2, 1, 1]):
        assert sum(results[r][1][li][3] for r in range(world)) == want_proved
        assert sum(results[r][1][li][4] for r in range(world)) == sum(results[r][1][li][5] for r in range(world))   # bytes sent == bytes received
    assert sum(results[r][1][0][4] for r in range(world)) > 0


# ---- a real guest across ranks: rank r proves shard r of ONE execution (bench.py's program workloads, bench/program_shard.py)
def _guest_shard_worker(rank, world, port, max_cycles, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import struct
    import pyoracle as orc
    from sp1_amd import shards
    from sp1_amd.machines import riscv_exec as X, riscv_trace as RT
    orc.set_threads(1)
    ex = X.Executor(X.guest_file("fibonacci.elf"), stdin=[struct.pack("<Q", 10_000)])
    prev = None
    for i in range(rank + 1):                       # the shards before this rank's run without keeping their events ...
        sh = ex.run_shard(max_cycles, record=i == rank)
        if i < rank:                                # ... but their public values chain into this rank's (prev_* fields)
            prev = X.execution_public_values(sh, prev)
    machine, tabs, publics = X.shard_tables(ex, sh, prev=prev)
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None) for a, i in machine]
    L, lsh, batch = 17, 12, 8
    prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, 1)
    ch = orc.Challenger()
    ch.observe(prep.commit)
    orc.set_gkr_sparse(True)
    proof = orc.shard_prove(host, RT.to_monty_np(publics), prep, L, lsh, batch, ch, 1, 5, 4)
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None) for a, i in machine]
    v_ch = orc.Challenger()
    v_ch.observe(prep.commit)
    from sp1_amd.machines import public_values as PVM
    ok = orc.shard_verify(shapes, prep.commit, proof, L, lsh, v_ch, 1, 5, 4, pv_program=PVM.verifier_program()) == 0
    merged = shards.gather_blobs({rank: proof})
    q.put((rank, ok, sh.index, sh.clk_start, sh.clk_end, sh.pc_start, sh.next_pc, sorted((k, len(v)) for k, v in merged.items())))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_prove_consecutive_shards_of_one_guest_execution_gloo():
    """The reference's fibonacci guest, world 2: rank r runs shards 0..r-1 with recording off and proves shard r (the oracle
    prover stands in for the GPU here); the proofs verify, the shards chain (clock and pc), and every rank ends up with both."""
    import __graft_entry__ as g
    g.build_hip()
    g.build_oracle()
    world, max_cycles = 2, 2000
    port = 29500 + (os.getpid() % 2000) + 911
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_guest_shard_worker, args=(r, world, port, max_cycles, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, i0, c0s, c0e, p0s, p0n, m0), (r1, ok1, i1, c1s, c1e, p1s, p1n, m1) = results
    assert ok0 and ok1 and (i0, i1) == (0, 1)
    assert c0s == 1 and c1s == c0e and p1s == p0n                 # shard 1 starts where shard 0 stopped
    assert m0 == m1 and [k for k, _ in m0] == [0, 1] and all(n > 10_000 for _, n in m0)


# ---- the work queue + event-driven compress tree (sp1_amd/scheduler.py): the reference's CoreWorker queue and CompressTree ----
# prove seconds of the 26 shards of the rsp block on one MI355X (profiles/r05_rsp_whole_block_final.json, `per_shard[].prove_ms`),
# in proof order: precompile | core | memory (compress.rs:L222-L233); a tree node = a compress proof of the recursion machine
# (23.6 ms: profiles/r05_bench_recursion.txt)
RSP_SHARD_MS = [91.35, 79.32, 80.1, 80.3, 63.34, 20.19, 20.7, 90.47, 100.9,
                114.8, 81.01, 78.97, 78.29, 80.52, 81.43, 81.22, 81.42, 88.08, 86.96, 87.48, 85.96, 83.36, 83.92, 37.56,
                69.16, 53.76]
JOIN_MS = 23.6


def _sched_worker(rank, world, port, n, arity, costs_ms, join_ms, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hashlib
    import time
    from sp1_amd import scheduler

    def leaf(i):
        if costs_ms:
            time.sleep(costs_ms[i] * 1e-3)
        return _shard_blob(i) * (1 + i % 3)                 # blobs of different lengths (incl. long ones)

    def combine(kids):
        if join_ms:
            time.sleep(join_ms * 1e-3)
        return hashlib.sha256(b"|".join(kids)).digest() + bytes([len(kids)])
    wq = scheduler.WorkQueue(n, arity, name="t")
    dist.barrier()
    t0 = time.perf_counter()
    root, stats = wq.run(leaf, combine)
    wall = time.perf_counter() - t0
    dist.barrier()
    q.put((rank, root, {k: stats[k] for k in ("leaves", "joins", "busy_s", "wait_s", "sent_bytes", "recv_bytes", "nodes", "store_ops")}, wall))
    dist.destroy_process_group()


def _run_scheduler(world, n, arity, costs_ms=None, join_ms=0.0):
    port = 33500 + (os.getpid() % 2000) + 13 * n + arity + 3 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sched_worker, args=(r, world, port, n, arity, costs_ms, join_ms, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("world,n,arity", [(2, 7, 2), (3, 5, 2), (2, 1, 2), (2, 2, 3), (8, 21, 2), (8, 9, 3), (8, 3, 2), (4, 0, 2)])
def test_work_queue_and_event_driven_tree_gloo(world, n, arity):
    """Every leaf is proved exactly once by whichever rank pulled it, every join combines ADJACENT ranges of proofs that exist, the
    recorded nodes form one tree over all leaves, and the root that reaches rank 0 is what recombining the recorded tree gives
    (children order = range order; blobs travel point to point between the ranks that made and used them)."""
    import hashlib
    import __graft_entry__ as g
    g.build_hip()
    from sp1_amd import scheduler
    results = _run_scheduler(world, n, arity)
    if n == 0:
        assert all(r[1] is None for r in results)
        return
    assert sorted(i for r in results for i in r[2]["leaves"]) == list(range(n))
    nodes = results[0][2]["nodes"]
    assert all(r[2]["nodes"] == nodes for r in results)                    # one tree, seen by everyone
    root_pid = scheduler.check_tree(n, nodes, arity)
    blob = {i: _shard_blob(i) * (1 + i % 3) for i in range(n)}
    for nd in nodes:
        blob[nd["pid"]] = hashlib.sha256(b"|".join(blob[c] for c in nd["children"])).digest() + bytes([len(nd["children"])])
    assert results[0][1] == blob[root_pid] and all(r[1] is None for r in results[1:])
    assert sum(len(r[2]["joins"]) for r in results) == len(nodes) and sum(r[2]["sent_bytes"] for r in results) == sum(r[2]["recv_bytes"] for r in results)


def test_work_queue_makespan_on_the_rsp_block_gloo():
    """World 8 over the 26 recorded shard times of the rsp block, a 23.6 ms compress proof per tree node. Held against
    (a) the same policy simulated with free transfers and a free control plane (scheduler.simulate): what the store round trips,
    the polling and the point-to-point transfers cost must stay under 12 % (measured here: 6-12 %); (b) the round-robin stripe with a barrier per tree
    level that rounds 1-5 used (scheduler.static_stripe_makespan, transfers free): the queue must beat it; (c) printed, not asserted:
    total work / 8 — no schedule of 26 leaves of ~80 ms on 8 ranks reaches it (the leaves alone are four rounds, and the proofs the
    ranks hold when the last round ends still need their joins)."""
    import __graft_entry__ as g
    g.build_hip()
    from sp1_amd import scheduler
    world, n = 8, len(RSP_SHARD_MS)
    work = (sum(RSP_SHARD_MS) + JOIN_MS * (n - 1)) * 1e-3
    ideal = scheduler.simulate([c * 1e-3 for c in RSP_SHARD_MS], world, JOIN_MS * 1e-3)
    for attempt in range(2):        # (8 ranks + their communication threads on this 8-core container: one join out of place is 6 % of the makespan)
        results = _run_scheduler(world, n, 2, RSP_SHARD_MS, JOIN_MS)
        makespan = max(r[3] for r in results)
        if makespan <= 1.12 * ideal:
            break
    nodes = results[0][2]["nodes"]
    scheduler.check_tree(n, nodes, 2)
    static = scheduler.static_stripe_makespan([c * 1e-3 for c in RSP_SHARD_MS], world, JOIN_MS * 1e-3)
    busy = sum(r[2]["busy_s"] for r in results)
    print("makespan %.3f s | same policy, free transfers %.3f | static stripe + level barriers %.3f | total work / 8 = %.3f | busy %.3f of %.3f rank-seconds"
          % (makespan, ideal, static, work / world, busy, world * makespan))
    for r in results:
        print("  rank %d: leaves %s joins %s busy %.3f wait %.3f store ops %d" % (r[0], r[2]["leaves"], r[2]["joins"], r[2]["busy_s"], r[2]["wait_s"], r[2]["store_ops"]))
    assert abs(busy - work) < 0.25                                         # every task ran once (sleep overshoot aside)
    assert makespan <= 1.12 * ideal
    assert makespan < 0.95 * static
    # the yardstick itself: one rank does everything in sequence; a single leaf needs no join
    assert abs(scheduler.simulate(RSP_SHARD_MS, 1, JOIN_MS) - (sum(RSP_SHARD_MS) + JOIN_MS * (n - 1))) < 1e-6 and scheduler.simulate([1.0], 4, 0.25) == 1.0
