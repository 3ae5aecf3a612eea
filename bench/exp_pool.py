#!/usr/bin/env python3
"""Experiment (GPU box): where does per-proof time vary? Direct proofs on the default stream / on a side stream / through a
1-slot pool, per-proof times, with the GPU's sclk + power sampled from OUR device's hwmon; then the same after forcing the
performance level to high (if sysfs lets us). usage: python bench/exp_pool.py [scale_log2]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bench"))
import torch  # noqa: E402
import bench as B  # noqa: E402
from core_shard import build_core_shard  # noqa: E402
from sp1_amd import api  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.cuda.set_device(0)
L, lsh = 22 - k, 21 - k
chips, meta = build_core_shard(B.CORE_AREA >> (2 * k), L)
pk = api.ProvingKey([c[3] for c in chips if c[3] is not None], L, lsh, 32)
torch.cuda.synchronize()


def direct(n, stream=None):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        if stream is None:
            pk.prove_shard(chips, [])
        else:
            with torch.cuda.stream(stream):
                pk.prove_shard(chips, [], stream=stream)
        out.append(round(1e3 * (time.perf_counter() - t0), 1))
    return out


def pooled(n_slots, n):
    pool = api.ProverPool(n_slots)
    for t in [pool.submit(pk, chips) for _ in range(n_slots)]:
        pool.wait(t)
    t0 = time.perf_counter()
    res = [pool.wait(t) for t in [pool.submit(pk, chips) for _ in range(n)]]
    dt = time.perf_counter() - t0
    pool.close()
    return round(1e3 * dt / n, 1), [round(r[1]["proving_ms"], 1) for r in res]


def phase(name, fn):
    with B.GpuSampler(0.02) as s:
        r = fn()
    print(name, r, s.summary(), flush=True)


def run_all(tag):
    phase(tag + " direct default stream", lambda: direct(8))
    side = torch.cuda.Stream()
    direct(1, side)
    phase(tag + " direct side stream", lambda: direct(8, side))
    phase(tag + " pool 1 slot", lambda: pooled(1, 8))
    phase(tag + " pool 2 slots", lambda: pooled(2, 8))
    phase(tag + " pool 3 slots", lambda: pooled(3, 9))


direct(2)
run_all("auto:")
s = B.GpuSampler()
lvl = None
if s.device:
    lvl = "/sys/bus/pci/devices/%s/power_dpm_force_performance_level" % s.device
try:
    print("perf level was:", open(lvl).read().strip())
    with open(lvl, "w") as f:
        f.write("high")
    print("perf level now:", open(lvl).read().strip())
    run_all("high:")
    with open(lvl, "w") as f:
        f.write("auto")
except Exception as e:
    print("cannot force the performance level:", repr(e))
