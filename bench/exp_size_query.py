import sys, time, ctypes as C
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import numpy as np, torch
from sp1_amd import api, _lib
from core_shard import build_core_shard
L, lsh = 22, 21
chips, meta = build_core_shard(3 << 27, L)
prep_tables = [c[3] for c in chips if c[3] is not None]
pk = api.ProvingKey(prep_tables, L, lsh, 32)
arr, keep = api._shard_chip_array(chips)
w = (C.c_uint32 * 1)()
n = C.c_size_t(0)
args = [pk.h, arr, len(chips), None, 0, w, 0]
for _ in range(3):
    t = time.perf_counter()
    st = api._L().sp1hip_prove_shard_with_pk(*args, None, C.byref(n), None)
    print("size query: %.3f ms (status %d, %d bytes)" % (1e3 * (time.perf_counter() - t), st, n.value))
t = time.perf_counter(); arr2, keep2 = api._shard_chip_array(chips); print("chip array marshalling: %.3f ms" % (1e3 * (time.perf_counter() - t)))
