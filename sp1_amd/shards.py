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
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_blobs(blobs):
    """All-gather {shard index: bytes} from every rank; returns the merged dict on every rank.

    Two fixed-shape collectives (counts/lengths, then zero-padded payloads), so it runs unchanged on
    RCCL (device tensors) and gloo (host tensors)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
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
