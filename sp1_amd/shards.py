"""Multi-GPU layout of the proving path: independent shards striped over ranks, one process per GPU.

The reference has no collective anywhere: multi-GPU = one prover process per device,
shards handed out by a queue (`CoreWorker`, /root/reference/crates/prover/src/worker/prover/core.rs:L255),
proofs moved as bincode blobs (`ArtifactClient`). The MI355X equivalent keeps that shape: rank r proves
shards r, r + W, r + 2W, …; nothing is exchanged while a shard is being proven; the only traffic is
the finished proof blobs (~1–2 MB each) travelling to the rank that needs them (the controller, or the
GPU proving the parent node of the recursion tree) — `gather_blobs` below, RCCL over xGMI when the
backend is "nccl", gloo in the CPU tests. Latency-bound, never bandwidth-bound.
"""
import torch
import torch.distributed as dist


def stripe(n_items, world_size, rank):
    """Indices of the shards rank `rank` proves (round-robin striping)."""
    return list(range(rank, n_items, world_size))


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def max_over_ranks(value):
    """Max of a python float over all ranks (used for the bench's max-over-ranks timing)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_blobs(blobs):
    """All-gather {shard index: bytes} from every rank; returns the merged dict on every rank.

    Two fixed-shape collectives (counts/lengths, then zero-padded payloads), so it runs unchanged on
    RCCL (device tensors) and gloo (host tensors)."""
    if not dist.is_initialized():
        return dict(blobs)
    world, dev = dist.get_world_size(), _device()
    items = sorted(blobs.items())
    n_local = torch.tensor([len(items)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local)
    max_n = max(int(c.item()) for c in counts)
    meta = torch.full((max(max_n, 1), 2), -1, dtype=torch.int64, device=dev)
    for k, (idx, b) in enumerate(items):
        meta[k, 0], meta[k, 1] = idx, len(b)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_len = max([int(m[:, 1].max().item()) for m in metas] + [1])
    payload = torch.zeros((max(max_n, 1), max_len), dtype=torch.uint8, device=dev)
    for k, (_, b) in enumerate(items):
        if len(b):                       # (torch.frombuffer rejects an empty buffer)
            payload[k, :len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    payloads = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(payloads, payload)
    out = {}
    for m, p in zip(metas, payloads):
        m, p = m.cpu(), p.cpu()
        for k in range(m.shape[0]):
            idx, ln = int(m[k, 0]), int(m[k, 1])
            if idx >= 0:
                out[idx] = bytes(p[k, :ln].numpy().tobytes())
    return out


def reduce_tree(leaf_blobs, n_leaves, combine, arity=2, on_level=None):
    """The recursion compress tree across ranks (`CompressTree::reduce_proofs`,
    /root/reference/crates/prover/src/worker/controller/compress.rs:L234-L420): adjacent ranges of proofs are batched
    `arity` at a time into a parent node until one proof is left. Nodes of every level are striped over the ranks like
    the leaves (node j of a level belongs to rank j mod W); a child that lives on another rank is SENT to the parent's
    rank — point-to-point (`batch_isend_irecv`: RCCL send/recv over xGMI under the "nccl" backend, gloo in the CPU
    tests), nothing is broadcast, and nothing moves while a node is being proven. This is the only inter-GPU traffic of
    the whole proving pipeline; it is latency-bound (~1.5 MB per proof).

    leaf_blobs: {leaf index: bytes} for the leaves THIS rank proved (`stripe`); n_leaves: global leaf count;
    combine(list_of_child_blobs) -> bytes proves a parent (a RecursionAir shard whose witness is the child proofs —
    the same `sp1hip_prove_shard` hot path with another machine description). A node with a single child is carried
    up unchanged. Returns the root blob on rank 0 and None elsewhere.
    on_level(level_index, stats): called on every rank after each level with this rank's {"nodes", "parents", "proved",
    "sent_bytes", "recv_bytes", "exchange_s", "prove_s"} (the bench's per-level table)."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    level = dict(leaf_blobs)
    n = int(n_leaves)
    if n == 0:
        return None
    level_index = 0
    while n > 1:
        n_parents = (n + arity - 1) // arity
        have = {}                       # child index -> blob, for the parents this rank owns
        sent = recvd = 0
        t_level = time.perf_counter()
        if world == 1:
            have = level
        else:
            dev = _device()
            # lengths of every node of the level: one small fixed-shape collective
            lens = torch.zeros(n, dtype=torch.int64, device=dev)
            for i, b in level.items():
                lens[i] = len(b)
            dist.all_reduce(lens, op=dist.ReduceOp.SUM)
            lens = lens.cpu().tolist()
            ops, recv_bufs, keep = [], {}, []
            for j in range(n_parents):
                owner = j % world
                for c in range(j * arity, min((j + 1) * arity, n)):
                    src = c % world
                    if src == owner:
                        if rank == owner:
                            have[c] = level[c]
                    elif lens[c] == 0:           # an empty blob needs no message
                        if rank == owner:
                            have[c] = b""
                    elif rank == src:
                        t = torch.frombuffer(bytearray(level[c]), dtype=torch.uint8).to(dev)
                        keep.append(t)
                        sent += len(level[c])
                        ops.append(dist.P2POp(dist.isend, t, owner))
                    elif rank == owner:
                        t = torch.empty(lens[c], dtype=torch.uint8, device=dev)
                        recv_bufs[c] = t
                        recvd += lens[c]
                        ops.append(dist.P2POp(dist.irecv, t, src))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for c, t in recv_bufs.items():
                have[c] = bytes(t.cpu().numpy().tobytes())
        t_prove = time.perf_counter()
        nxt, proved = {}, 0
        for j in range(rank, n_parents, world):
            kids = [have[c] for c in range(j * arity, min((j + 1) * arity, n))]
            if len(kids) == 1:
                nxt[j] = kids[0]
            else:
                nxt[j] = combine(kids)
                proved += 1
        if on_level is not None:
            on_level(level_index, {"nodes": n, "parents": n_parents, "proved": proved, "sent_bytes": sent, "recv_bytes": recvd,
                                   "exchange_s": t_prove - t_level, "prove_s": time.perf_counter() - t_prove})
        level, n = nxt, n_parents
        level_index += 1
    return level.get(0) if rank == 0 else None


def recursion_combine(prove, counts, seed=0):
    """A real `combine` for `reduce_tree`: the parent of a list of child proofs is a shard proof over the reference's
    recursion compress machine (sp1_amd/machines/recursion.py) that COMMITS to its children — the eight words of its
    public-values digest (RecursionPublicValues::digest, constrained by the PublicValues chip) are derived from the
    children's bytes. The recursion *verifier program* that would check the children inside the VM is Rust and out of
    scope; what travels through the tree and gets proven at every node is a genuine ShardProof of that
    machine, produced by `prove(tables, public_values) -> bytes` (sp1hip_prove_shard behind `ProvingKey.prove_shard`
    on a GPU rank; the CPU tests plug the oracle prover in).

    counts: rows per chip of the parent's (random straight-line) recursion program, see recursion_trace.generate."""
    import hashlib

    from .machines import recursion_trace

    def combine(children):
        h = hashlib.sha256()
        for c in children:
            h.update(len(c).to_bytes(8, "little"))
            h.update(c)
        d = h.digest()
        digest = [int.from_bytes(d[4 * i:4 * i + 4], "little") % recursion_trace.P for i in range(8)]
        tables, publics = recursion_trace.generate(counts, seed=seed + len(children), digest=digest)
        return prove(tables, publics)

    return combine
