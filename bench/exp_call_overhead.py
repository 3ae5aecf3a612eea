"""Host time around one proof as the bench loop sees it: marshalling, the C call, the copy out (run on the GPU box)."""
import os, sys, time, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np, torch
from sp1_amd import api, _lib
from core_shard import build_core_shard
L, lsh = 22, 21
chips, meta = build_core_shard(3 << 27, L)
jp = api.JaggedProver(L, lsh, 32, 2)
prep_commit, prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
for rep in range(5):
    ch = api.DuplexChallenger(); ch.observe(prep_commit)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); arr, keep = api._shard_chip_array(chips); t1 = time.perf_counter()
    proof = api.prove_shard(chips, [], prep, L, lsh, 32, ch); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("marshal %.3f ms | prove_shard (incl. its own marshalling + copy out) %.3f ms | sync after %.3f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)))
