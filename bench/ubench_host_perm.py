"""Host permutation latency on this machine: the transcript's form (AVX-512 where available) against the two scalar
forms, ns per permutation over a batch of independent states (the transcript's permutations are dependent, but one
permutation is already a single dependency chain, so the batch rate is the chain rate)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "sp1_amd", "lib", "libsp1hip.so"))
lib.sp1hip_poseidon2_permute_host.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
P = 0x7F000001
n = 400000
base = np.random.default_rng(1).integers(0, P, size=(n, 16), dtype=np.uint32)
res = {}
for form, name in ((0, "transcript"), (1, "scalar_int"), (2, "scalar_f64")):
    s = base.copy()
    lib.sp1hip_poseidon2_permute_host(s.ctypes.data, 1000, form)
    t = time.perf_counter()
    lib.sp1hip_poseidon2_permute_host(s.ctypes.data, n, form)
    res[name] = (time.perf_counter() - t) / n * 1e9
print("host permutation ns: " + ", ".join("%s %.0f" % kv for kv in res.items()) +
      "; vectorised: %d" % lib.sp1hip_host_permutation_is_vectorised())
