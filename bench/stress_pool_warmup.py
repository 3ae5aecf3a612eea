#!/usr/bin/env python3
"""Stress of the pattern bench.py's in_flight() phases start with: trim the arena, create a pool of n slots, prove n shards at
once on a cold arena (every buffer a fresh hipMalloc, the slots' helper streams created on first use), destroy the pool — repeated.
usage: python bench/stress_pool_warmup.py [iterations] [scale_log2]   (SP1HIP_WAIT_TIMEOUT_S shortens the hand-over time-outs)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bench"))
import faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("STRESS_WATCHDOG", "400")), exit=True)
import torch
from sp1_amd import api
import core_real

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.set_device(0)
chips, meta = core_real.build_real_shard(scale=1.0 / (1 << (2 * k)))
L, lsh = max(22 - k, 17), 21 - k
pk = api.ProvingKey([c[3] for c in chips if c[3] is not None], L, lsh, 32)
want = pk.prove_shard(chips, [])
torch.cuda.synchronize()
ok = 0
for it in range(iters):
    for n in (2, 3, 4):
        released = C.c_size_t()
        api.check(api._L().sp1hip_mem_trim(C.byref(released)))
        pool = api.ProverPool(n)
        t0 = time.perf_counter()
        try:
            for t in [pool.submit(pk, chips) for _ in range(n)]:
                assert pool.wait(t)[0] == want, "a pool proof differs from the direct one"
            for t in [pool.submit(pk, chips) for _ in range(2 * n)]:
                assert pool.wait(t)[0] == want, "a pool proof differs from the direct one"
        except Exception as e:                                  # noqa: BLE001
            print("FAILED at iteration %d, %d slots after %.1f s: %s" % (it, n, time.perf_counter() - t0, e), flush=True)
            sys.exit(1)
        pool.close()
        ok += 3 * n
print("stress OK: %d pool proofs over %d iterations" % (ok, iters))
