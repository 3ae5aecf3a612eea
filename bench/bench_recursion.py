#!/usr/bin/env python3
"""One recursion (compress) shard with the REAL machine at the REAL shape: the eight chips of
RecursionAir::compress_machine() (sp1_amd/machines/recursion.py, pinned by the reference's own proof) with the table
heights of the reference's compress proof in shrink_input.bin (8.9e7 cells, max_log_row_count 21, stacking height 2^20,
blowup 4, 124 queries, 16-bit PoW), satisfying traces from sp1_amd/machines/recursion_trace.py, resident in HBM.

  python bench/bench_recursion.py [--scale-log2 K] [--repeat R]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sp1_amd import api  # noqa: E402
from sp1_amd.machines import recursion as R, recursion_trace as RT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale-log2", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=4)
    ap.add_argument("--stages", action="store_true")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    k = args.scale_log2
    L, lsh = 21 - k, 20 - k
    counts = {n: max(h >> k, 8) for n, h in RT.REFERENCE_COMPRESS_HEIGHTS.items()}
    t0 = time.perf_counter()
    tabs, pv = RT.generate(counts, seed=1)
    m = R.compress_machine()
    area = sum(p.size + mm.size for p, mm in tabs.values())
    print("generated %.3e cells in %.1f s" % (area, time.perf_counter() - t0), file=sys.stderr)
    dev = [(a, i, api.ColMajor.from_row_major_host(tabs[a.name][1]), api.ColMajor.from_row_major_host(tabs[a.name][0])) for a, i in m]
    jp = api.JaggedProver(L, lsh, 32, 2)
    commit, prep = jp.commit_multilinears([d[3] for d in dev])
    names = ("ntt_pass0", "ntt_pass1", "ntt_pass2", "leaf_hash", "compress", "gkr_first_layer", "gkr_transition", "gkr_pass_sum", "gkr_pass_fold_sum", "gkr_pass_fold",
             "gkr_openings", "zerocheck_round", "zerocheck_fix", "jagged_round0_sum", "jagged_fold0_sum",
             "jagged_fold_sum", "jagged_batch_evals")
    for rep in range(args.repeat):
        last = rep == args.repeat - 1
        api.check(api._L().sp1hip_timers_enable(1 if last else 0))
        api.check(api._L().sp1hip_timers_reset())
        ch = api.DuplexChallenger()
        ch.observe(commit)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proof = api.prove_shard(dev, pv, prep, L, lsh, 32, ch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        out = {"workload": "recursion compress shard, reference shape >> %d" % k, "cells": int(area), "prove_shard_ms": round(ms, 2),
               "cells_per_s": round(area / ms * 1e3), "proof_bytes": len(proof)}
        if last:
            t = {}
            for name in names:
                cnt, tms = C.c_uint64(), C.c_double()
                api.check(api._L().sp1hip_timers_read(name.encode(), C.byref(cnt), C.byref(tms)))
                if cnt.value:
                    t[name + "_ms"] = round(tms.value, 3)
            out["kernel_ms_with_timers_on"] = t
        print(json.dumps(out), flush=True)
    api.check(api._L().sp1hip_timers_enable(0))
    if args.stages:
        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            return r, (time.perf_counter() - t0) * 1e3
        ch = api.DuplexChallenger()
        ch.observe(commit)
        ch.observe(pv)
        (mc, sd), t_commit = timed(lambda: jp.commit_multilinears([d[2] for d in dev]))
        ch.observe(mc)
        gblob, t_gkr = timed(lambda: api.logup_gkr([(d[1], d[2], d[3]) for d in dev], L, ch))
        zeta, opened = api.parse_logup_gkr_proof(gblob)
        alpha, gkr_b = ch.sample_ext_element(), ch.sample_ext_element()
        ops = np.concatenate([np.concatenate([mm] + ([p] if p is not None else [])) for _, mm, p in opened])
        zchips = [api.ZerocheckChip(d[0], d[2], d[3]) for d in dev]
        zblob, t_zc = timed(lambda: api.zerocheck(zchips, L, zeta, ops, alpha, gkr_b, pv, ch))
        print(json.dumps({"stages_ms": {"commit": round(t_commit, 2), "logup_gkr": round(t_gkr, 2), "zerocheck": round(t_zc, 2)}}), flush=True)


if __name__ == "__main__":
    main()
