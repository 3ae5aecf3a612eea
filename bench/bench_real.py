#!/usr/bin/env python3
"""First timing of the real-chip core shard: python bench/bench_real.py [scale_log2] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bench"))
import ctypes as C
import faulthandler
faulthandler.dump_traceback_later(int(os.environ.get('BENCH_REAL_WATCHDOG', '120')), exit=True)
import torch
from sp1_amd import api
import core_real
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
t0 = time.perf_counter()
chips, meta = core_real.build_real_shard(scale=1.0 / (1 << (2 * k)))
torch.cuda.synchronize()
print("built in %.1fs: area %.3e real %.3e chips %d" % (time.perf_counter() - t0, meta["area_cells"], meta["real_area_cells"], meta["chips"]), file=sys.stderr)
L, lsh = 22 - k, 21 - k
L = max(L, 17)
jp = api.JaggedProver(L, lsh, 32, 2)
commit, prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
lib = api._L()
TIMERS = ("ntt_pass0", "ntt_pass1", "ntt_pass2", "leaf_hash", "compress", "gkr_first_layer", "gkr_transition", "gkr_pass_sum", "gkr_pass_fold_sum", "gkr_pass_fold",
          "gkr_openings", "zerocheck_round", "zerocheck_fix", "jagged_round0_sum", "jagged_fold0_sum", "jagged_fold_sum", "jagged_batch_evals")
def step():
    ch = api.DuplexChallenger(); ch.observe(commit)
    return api.prove_shard(chips, [], prep, L, lsh, 32, ch), ch
for _ in range(2): step()
torch.cuda.synchronize()
api.check(lib.sp1hip_timers_reset()); api.check(lib.sp1hip_timers_enable(1))
t0 = time.perf_counter()
for _ in range(steps): proof, ch = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
api.check(lib.sp1hip_timers_enable(0))
tl = {}
for name in TIMERS:
    n, ms = C.c_uint64(), C.c_double()
    api.check(lib.sp1hip_timers_read(name.encode(), C.byref(n), C.byref(ms)))
    if n.value: tl[name] = round(ms.value / steps, 3)
print(json.dumps({"ms_per_proof": 1e3 * dt, "cells_per_s": meta["area_cells"] / dt, "area": meta["area_cells"], "real_area": meta["real_area_cells"],
                  "proof_bytes": len(proof), "timers_ms": tl, "constraints": meta["constraints"], "interactions": meta["interactions"],
                  "first_layer_entries": meta["first_layer_entries"]}))
