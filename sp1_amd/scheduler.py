"""Shards of one proof over the GPUs of a node in the reference's shape: a WORK QUEUE and a compress tree that joins
ADJACENT RANGES AS PROOFS ARRIVE — no static assignment, no per-level barrier.

The reference (one prover process per device, no collective anywhere — SURVEY §2.4):
  * workers pull the next task from a queue behind a semaphore       /root/reference/crates/prover/src/worker/prover/core.rs:L255,
                                                                      crates/hypercube/src/prover/permits.rs:L36-L66
  * `CompressTree` keeps the finished proofs as ranges in a BTreeMap; a proof that arrives is merged with the range that ends
    where it starts and / or the one that starts where it ends (`sibling`), a group that reaches the batch size — or completes
    the full range — becomes a reduce task, the remainder goes back into the map (`reduce_proofs`)
                                                                      crates/prover/src/worker/controller/compress.rs:L234-L470
  * leaves are ordered  precompile | deferred | core | memory  so that every proof has a neighbour (compress.rs:L222-L233).

The MI355X form keeps one process per GPU and `torch.distributed`, and splits control from data:
  * control plane = the process group's TCPStore: an atomic counter hands out leaves (`add`), the tree state (the range map, the
    outboxes, the finished nodes) is one pickled value behind a store lock — a few hundred small operations per proof;
  * data plane = point-to-point `isend` / `irecv` of finished proof blobs (~1.5 MB) on the process group — RCCL over xGMI under
    the "nccl" backend, gloo in the CPU tests. Reduce tasks sit in a queue like the leaves; the rank that takes one asks the
    children's owners for the blobs it does not hold — a send request in their outbox, which a communication thread of every rank
    serves WHILE the rank proves: a join never waits for a neighbour's proof to finish, there is no cycle of blocked ranks, and
    nothing is broadcast.

`run()` returns the root blob on rank 0 and, on every rank, what it did (tasks, busy / waiting seconds, bytes moved) plus the
tree that was actually built — its shape depends on the order proofs finish in, which is the point: with shard times spread
20-117 ms (profiles/r05_rsp_whole_block_final.json) a fixed round-robin stripe and a barrier per level leave GPUs idle.
"""
import pickle
import time

import torch
import torch.distributed as dist


class _LocalStore:
    """The store API this module uses, in-process (world size 1, or no process group at all)."""

    def __init__(self):
        self.d = {}

    def add(self, key, n):
        self.d[key] = int(self.d.get(key, 0)) + n
        return self.d[key]

    def set(self, key, value):
        self.d[key] = value

    def get(self, key):
        return self.d[key]


def _default_store():
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return _LocalStore()
    return dist.distributed_c10d._get_default_store()


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


class WorkQueue:
    """One run of the scheduler. Every rank builds it with the same arguments and calls `run`."""

    def __init__(self, n_leaves, arity=2, name="sp1", store=None, poll_s=0.001):
        assert arity >= 2
        self.n, self.arity, self.poll_s = int(n_leaves), int(arity), poll_s
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.store = store if store is not None else _default_store()
        self.k = lambda s: "%s/%s" % (name, s)
        self.blobs, self.sends, self.serviced = {}, [], 0                    # proofs held here, sends in flight, outbox entries served
        self._notify = []
        self.device = _device() if dist.is_initialized() else torch.device("cpu")   # (the current device is per thread: fixed here)
        self.stats = {"leaves": [], "joins": [], "busy_s": 0.0, "wait_s": 0.0, "sent_bytes": 0, "recv_bytes": 0, "store_ops": 0}
        if self.rank == 0:
            self._save({"ranges": {}, "pending": self.n, "leaves_done": 0, "next_pid": self.n, "outbox": {r: [] for r in range(self.world)},
                        "root": None, "nodes": [], "tasks": []})
            self.store.set(self.k("ready"), b"1")
        self._barrier_on_key("ready")

    # -- the store: a lock, the pickled state
    def _barrier_on_key(self, key):
        while True:
            try:
                self.store.get(self.k(key))
                return
            except Exception:
                time.sleep(self.poll_s)

    def _lock(self):
        backoff = self.poll_s
        while self.store.add(self.k("lock"), 1) != 1:
            self.store.add(self.k("lock"), -1)
            time.sleep(backoff)
            backoff = min(2 * backoff, 0.004)
        self.stats["store_ops"] += 1

    def _unlock(self):
        self.store.add(self.k("lock"), -1)

    def _load(self):
        self.stats["store_ops"] += 1
        return pickle.loads(bytes(self.store.get(self.k("state"))))

    def _save(self, st):
        self.store.set(self.k("state"), pickle.dumps(st))

    # -- `CompressTree` (compress.rs:L236-L298): ranges keyed by their start; a proof = (pid, start, end, owner, length)
    @staticmethod
    def _sibling(ranges, start, end):
        left = next((s for s, r in ranges.items() if r["end"] == start), None)
        right = end if end in ranges else None
        return (ranges.pop(left) if left is not None else None), (ranges.pop(right) if right is not None else None)

    def _complete(self, st, start, end):
        return st["pending"] == 0 and not st["ranges"] and (start, end) == (0, self.n)

    def _arrived(self, proof, is_leaf):
        """A finished proof enters the tree (`reduce_proofs`, compress.rs:L418-L470, with the store lock held). A group that is
        full — or completes the range — goes onto the task queue."""
        self._lock()
        try:
            st = self._load()
            st["pending"] -= 1
            st["leaves_done"] += int(is_leaf)
            pid, start, end, owner, length = proof
            task = None
            if self._complete(st, start, end):
                st["root"] = proof
                if owner != 0:
                    st["outbox"][owner].append((pid, 0))
                    self._notify.append(owner)
            else:
                left, right = self._sibling(st["ranges"], start, end)
                if left is None and right is None:
                    st["ranges"][start] = {"end": end, "proofs": [proof]}
                else:
                    group = (left["proofs"] if left else []) + [proof] + (right["proofs"] if right else [])
                    rest = group[self.arity:]                             # `split_off`: the remainder goes back into the map
                    group = group[:self.arity]
                    if rest:
                        st["ranges"][rest[0][1]] = {"end": rest[-1][2], "proofs": rest}
                    g0, g1 = group[0][1], group[-1][2]
                    if len(group) == self.arity or self._complete(st, g0, g1):
                        st["pending"] += 1
                        st["tasks"].append(group)                          # a reduce task: any free rank takes it (the worker queue)
                        task = group
                    else:
                        st["ranges"][g0] = {"end": g1, "proofs": group}
            self._save(st)
            for r in self._notify:                                        # (after the state: a rank that sees the count finds the request)
                self.store.add(self.k("outbox_n/%d" % r), 1)
            self._notify.clear()
            if task is not None:
                self.store.add(self.k("tasks_n"), 1)
            return task
        finally:
            self._unlock()

    def _claim_join(self):
        """Take a reduce task off the queue — one this rank holds a child of, if there is such a one, else the oldest — and ask the
        other children's owners for their blobs. None when the queue is empty."""
        if self.store.add(self.k("tasks_n"), 0) <= 0:
            return None
        self._lock()
        try:
            st = self._load()
            if not st["tasks"]:
                return None
            k = next((j for j, g in enumerate(st["tasks"]) if any(c[3] == self.rank for c in g)), 0)
            group = st["tasks"].pop(k)
            for cpid, _, _, cowner, _ in group:
                if cowner != self.rank:
                    st["outbox"][cowner].append((cpid, self.rank))
                    self._notify.append(cowner)
            self._save(st)
            self.store.add(self.k("tasks_n"), -1)
            for r in self._notify:
                self.store.add(self.k("outbox_n/%d" % r), 1)
            self._notify.clear()
            return group
        finally:
            self._unlock()

    # -- the data plane: one communication thread per rank issues every point-to-point call
    def _comm_loop(self):
        """Serve this rank's outbox and complete the receives the worker asked for — while the worker proves (its GPU call has
        released the GIL). A blob leaves as soon as somebody needs it, not when its owner next comes up for air: a join never
        waits for a neighbour's 100 ms proof to finish. The sender marks `sent/<pid>` in the store after its `isend`; the receiver
        posts the matching `irecv` when it sees the mark, so neither side's transfer sits unmatched (an unmatched RCCL receive
        is a kernel spinning on the GPU) and `wait()` returns promptly (gloo's `Work.is_completed()` stays false until then)."""
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        while not self._stop:
            try:
                # (one small counter per rank says how many requests its outbox holds: the state is only loaded when there is news)
                mine = self._load()["outbox"][self.rank] if self.store.add(self.k("outbox_n/%d" % self.rank), 0) > self.serviced else ()
                while self.serviced < len(mine):
                    pid, dst = mine[self.serviced]
                    with self._mu:
                        blob = self.blobs.pop(pid)                        # a proof is consumed exactly once
                    if len(blob):
                        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device)
                        self.sends.append((dist.isend(t, dst, tag=pid % 32768), t))
                        self.store.add(self.k("sent/%d" % pid), 1)
                        self.stats["sent_bytes"] += len(blob)
                    self.serviced += 1
                with self._mu:
                    wanted = list(self._wanted.items())
                for pid, (owner, length) in wanted:
                    if self.store.add(self.k("sent/%d" % pid), 0) > 0:
                        t = torch.empty(length, dtype=torch.uint8, device=self.device)
                        dist.irecv(t, owner, tag=pid % 32768).wait()
                        with self._mu:
                            self.blobs[pid] = bytes(t.cpu().numpy().tobytes())
                            del self._wanted[pid]
                        self.stats["recv_bytes"] += length
            except Exception as e:                                        # surfaces in the worker's loop
                self._comm_error = e
                return
            time.sleep(self.poll_s)

    def _want(self, children):
        """Ask the communication thread for the children held elsewhere."""
        with self._mu:
            for pid, _, _, owner, length in children:
                if owner != self.rank and pid not in self.blobs:
                    if length == 0:
                        self.blobs[pid] = b""
                    else:
                        self._wanted[pid] = (owner, length)

    def _have(self, children):
        if self._comm_error is not None:
            raise RuntimeError("scheduler communication thread failed") from self._comm_error
        with self._mu:
            return all(c[0] in self.blobs for c in children)

    # -- a worker
    def run(self, prove_leaf, combine):
        """prove_leaf(i) -> bytes; combine([child bytes in range order]) -> bytes. Returns (root blob on rank 0 | None, stats).

        A rank's loop: take a reduce task if the queue has one (preferring a group it holds a child of; the other children come
        from their owners' communication threads within a millisecond or two); else claim the next leaf; else wait."""
        import threading
        t_start = time.perf_counter()
        if self.n == 0:
            self.stats["nodes"], self.stats["wall_s"] = [], 0.0
            return None, self.stats
        self._mu, self._wanted, self._stop, self._comm_error = threading.Lock(), {}, False, None
        comm = None
        if self.world > 1:
            comm = threading.Thread(target=self._comm_loop, daemon=True)
            comm.start()
        exhausted, st = False, None
        while True:
            task = self._claim_join()                                     # joins first: they are what the end of the run waits for
            if task is not None:
                self._want(task)
                t0 = time.perf_counter()
                while not self._have(task):                               # a millisecond or two: the owners' communication threads answer at once
                    time.sleep(self.poll_s / 2)
                self.stats["wait_s"] += time.perf_counter() - t0
                with self._mu:
                    kids = [self.blobs.pop(c[0]) for c in task]
                t0 = time.perf_counter()
                blob = combine(kids)
                self.stats["busy_s"] += time.perf_counter() - t0
                self._lock()
                try:
                    st = self._load()
                    pid = st["next_pid"]
                    st["next_pid"] += 1
                    st["nodes"].append({"pid": pid, "range": (task[0][1], task[-1][2]), "children": [c[0] for c in task], "rank": self.rank})
                    self._save(st)
                finally:
                    self._unlock()
                with self._mu:
                    self.blobs[pid] = blob
                self.stats["joins"].append((task[0][1], task[-1][2]))
                self._arrived((pid, task[0][1], task[-1][2], self.rank, len(blob)), False)
                continue
            if not exhausted:
                i = self.store.add(self.k("next_leaf"), 1) - 1
                self.stats["store_ops"] += 1
                if i < self.n:
                    t0 = time.perf_counter()
                    blob = prove_leaf(i)
                    self.stats["busy_s"] += time.perf_counter() - t0
                    with self._mu:
                        self.blobs[i] = blob
                    self.stats["leaves"].append(i)
                    self._arrived((i, i, i + 1, self.rank, len(blob)), True)
                    continue
                exhausted = True
            # nothing to run: wait for a task to appear, or for the root to exist, be where it belongs, and the outbox to be empty
            st = self._load()
            if st["root"] is not None and self.serviced == len(st["outbox"][self.rank]):
                root = st["root"]
                if self.rank != 0 or root[3] == 0:
                    break
                self._want([root])                                         # the root was made elsewhere: it travels to rank 0
                if self._have([root]):
                    break
            t0 = time.perf_counter()
            time.sleep(self.poll_s)
            self.stats["wait_s"] += time.perf_counter() - t0
        self._stop = True
        if comm is not None:
            comm.join()
        for r, _ in self.sends:
            r.wait()
        self.stats["nodes"], self.stats["wall_s"] = st["nodes"], time.perf_counter() - t_start
        return (self.blobs.get(st["root"][0]) if self.rank == 0 else None), self.stats


def simulate(costs, world, join_cost, arity=2):
    """The same policy as a discrete-event simulation with free transfers and a free control plane: the yardstick a run is held
    against (tests/test_multirank.py). A free rank takes the oldest reduce task, else the next leaf; a completion that fills a
    group of `arity` adjacent proofs (or completes the range) queues its join. Returns the makespan."""
    import heapq
    n = len(costs)
    if n == 0:
        return 0.0
    ranges, tasks, running, free = {}, [], [], list(range(world))
    nxt, pending, t = 0, n, 0.0

    def dispatch():
        nonlocal nxt
        while free and (tasks or nxt < n):
            r = free.pop()
            if tasks:
                s, e = tasks.pop(0)
                heapq.heappush(running, (t + join_cost, r, (s, e)))
            else:
                heapq.heappush(running, (t + costs[nxt], r, (nxt, nxt + 1)))
                nxt += 1
    dispatch()
    while running:
        t, r, (s, e) = heapq.heappop(running)
        free.append(r)
        pending -= 1
        if pending == 0 and not ranges and (s, e) == (0, n):
            return t
        left = next((a for a, grp in ranges.items() if grp[-1][1] == s), None)
        group = (ranges.pop(left) if left is not None else []) + [(s, e)] + (ranges.pop(e) if e in ranges else [])
        rest, group = group[arity:], group[:arity]
        if rest:
            ranges[rest[0][0]] = rest
        if len(group) == arity or (len(group) > 1 and pending == 0 and not ranges and (group[0][0], group[-1][1]) == (0, n)):
            pending += 1
            tasks.append((group[0][0], group[-1][1]))
        else:
            ranges[group[0][0]] = group
        dispatch()
    return t


def check_tree(n_leaves, nodes, arity):
    """The nodes a run recorded form one tree over the leaves 0 .. n - 1: every node joins 2 .. arity ADJACENT ranges, every proof
    is consumed exactly once, the last node covers everything. Returns the root's pid (n_leaves == 1: the leaf itself)."""
    rng = {i: (i, i + 1) for i in range(n_leaves)}
    used = set()
    for nd in nodes:
        kids = nd["children"]
        assert 2 <= len(kids) <= arity and not (set(kids) & used), nd
        spans = [rng[k] for k in kids]
        assert all(spans[j][1] == spans[j + 1][0] for j in range(len(spans) - 1)), ("children are not adjacent", nd)
        assert (spans[0][0], spans[-1][1]) == tuple(nd["range"]), nd
        used |= set(kids)
        rng[nd["pid"]] = tuple(nd["range"])
    roots = [p for p in rng if p not in used]
    assert len(roots) == 1 and rng[roots[0]] == (0, n_leaves), roots
    return roots[0]


def static_stripe_makespan(costs, world, join_cost, arity=2):
    """What the round-robin stripe + level-synchronous tree of rounds 1-5 (shards.stripe / shards.reduce_tree) takes on the same
    per-leaf costs, transfers free: max over ranks of their leaves, then per level the slowest rank's share of the parents."""
    t = max(sum(costs[r::world]) for r in range(world))
    n = len(costs)
    while n > 1:
        parents = (n + arity - 1) // arity
        proved = [sum(1 for j in range(r, parents, world) if min((j + 1) * arity, n) - j * arity > 1) for r in range(world)]
        t += max(proved) * join_cost
        n = parents
    return t
