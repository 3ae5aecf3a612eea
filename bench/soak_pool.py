"""Soak (GPU box): N proofs of a core shard through a 3-slot pool and N/4 on the direct path, every proof compared
byte for byte with the first one — the hand-over protocols (rs_finish, direct host rows, mailbox publishes) run ~350 times
per proof. usage: python bench/soak_pool.py [n_proofs] [real|core]     (real: the rv64im machine, bench.py's workload — default;
core: the synthetic core-shaped shard of rounds 1-3)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import torch
from sp1_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
kind = sys.argv[2] if len(sys.argv) > 2 else "real"
L, lsh = 22, 21
torch.cuda.set_device(0)
if kind == "real":
    import core_real
    chips, meta = core_real.build_real_shard()
else:
    from core_shard import build_core_shard
    chips, meta = build_core_shard(3 << 27, L)
pk = api.ProvingKey([c[3] for c in chips if c[3] is not None], L, lsh, 32)
want = pk.prove_shard(chips, [])
bad = 0
t0 = time.perf_counter()
pool = api.ProverPool(3)
tickets = [pool.submit(pk, chips) for _ in range(n)]
for t in tickets:
    bad += pool.wait(t)[0] != want
pool.close()
t1 = time.perf_counter()
for _ in range(n // 4):
    bad += pk.prove_shard(chips, []) != want
t2 = time.perf_counter()
print("soak: %d pooled proofs in %.1f s (%.1f ms each), %d direct in %.1f s (%.1f ms each), %d differ" %
      (n, t1 - t0, 1e3 * (t1 - t0) / n, n // 4, t2 - t1, 1e3 * (t2 - t1) / max(n // 4, 1), bad))
sys.exit(1 if bad else 0)
