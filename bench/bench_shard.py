#!/usr/bin/env python3
"""Stage timings of the implemented hot path at CORE-shard scale (A = 2^28 + 2^27 cells,
stacking height 2^21, 32 columns per batch, log_blowup 2, 124 queries, 16 PoW bits).

Not the driver's bench (that is ../bench.py, BASELINE config 2). This script proves one synthetic shard
resident in HBM end to end with sp1hip_prove_shard (commit -> LogUp-GKR -> zerocheck -> jagged evaluation
proof = a complete bincode(ShardProof); the chips have degree-3 constraints and balanced lookups, so the
proof is a valid one) and then times the four stages separately on the same inputs.

  python bench/bench_shard.py [--scale-log2 K]   (area = (2^28 + 2^27) >> K; default K = 0)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bench"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sp1_amd import api  # noqa: E402
from sp1_amd.air import AirProgram  # noqa: E402,F401
from synthetic_shard import build_shard  # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def read_timers(names):
    out = {}
    for name in names:
        cnt, ms = C.c_uint64(), C.c_double()
        api.check(api._L().sp1hip_timers_read(name.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value:
            out[name + "_ms"] = round(ms.value, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale-log2", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--core-shaped", action="store_true", help="the bench.py workload (bench/core_shard.py) instead of the round-1 one")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    L = 22 - args.scale_log2            # max_log_row_count
    lsh = 21 - args.scale_log2          # log stacking height
    area_target = ((1 << 28) + (1 << 27)) >> (2 * args.scale_log2) if args.scale_log2 else (1 << 28) + (1 << 27)
    if args.core_shaped:
        from core_shard import build_core_shard
        chips, meta = build_core_shard(area_target, L)
        area, shapes = meta["area_cells"], "core-shaped"
        prep_tables = [c[3] for c in chips if c[3] is not None]
    else:
        chips, prep_prep, shapes, area = build_shard(L, lsh, area_target)
        prep_tables = [prep_prep]
    n_int = sum(c[1].num_interactions for c in chips)
    print("chips: %d (%d interactions), shapes %s, area = %.3e cells" % (len(chips), n_int, shapes, area), file=sys.stderr)

    jp = api.JaggedProver(L, lsh, 32, 2)
    prep_commit, prep_data = jp.commit_multilinears(prep_tables)
    res = {"area_cells": area, "max_log_row_count": L, "log_stacking_height": lsh, "chips": len(chips), "interactions": n_int}
    stage_timers = ("ntt_pass0", "ntt_pass1", "ntt_pass2", "leaf_hash", "compress", "gkr_first_layer", "gkr_transition",
                    "gkr_pass_sum", "gkr_pass_fold_sum", "gkr_pass_fold", "gkr_openings", "zerocheck_round", "zerocheck_fix", "jagged_round0_sum", "jagged_fold0_sum",
                    "jagged_fold_sum", "jagged_batch_evals")
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for rep in range(args.repeat):
        ch = api.DuplexChallenger()
        ch.observe(prep_commit)
        api.check(api._L().sp1hip_timers_enable(1 if rep == args.repeat - 1 else 0))
        api.check(api._L().sp1hip_timers_reset())
        proof, t_total = timed(lambda: api.prove_shard(chips, [], prep_data, L, lsh, 32, ch))
        out = dict(res, prove_shard_ms=round(t_total, 2), shard_proof_bytes=len(proof),
                   cells_per_s=round(area / (t_total * 1e-3)))
        if rep == 0:      # what one in-flight proof holds on top of its inputs (the arena keeps it cached afterwards)
            out["hbm_working_set_gb"] = round((free0 - torch.cuda.mem_get_info()[0]) / 1e9, 2)
        if rep == args.repeat - 1:
            out["kernel_ms_with_timers_on"] = read_timers(stage_timers)
        print(json.dumps(out), flush=True)
    api.check(api._L().sp1hip_timers_enable(0))

    # stage by stage (same inputs), for the per-stage table in DESIGN.md
    ch = api.DuplexChallenger()
    ch.observe(prep_commit)
    (commit, sd), t_commit = timed(lambda: jp.commit_multilinears([c[2] for c in chips]))
    ch.observe(commit)
    gk = [(c[1], c[2], c[3]) for c in chips]
    gblob, t_gkr = timed(lambda: api.logup_gkr(gk, L, ch))
    zeta, opened = api.parse_logup_gkr_proof(gblob)
    alpha, gkr_b = ch.sample_ext_element(), ch.sample_ext_element()
    ops = np.concatenate([np.concatenate([m] + ([p] if p is not None else [])) for _, m, p in opened])
    zchips = [api.ZerocheckChip(c[0], c[2], c[3]) for c in chips]
    zblob, t_zc = timed(lambda: api.zerocheck(zchips, L, zeta, ops, alpha, gkr_b, [], ch))
    z_row, chip_evals = api.parse_zerocheck_proof(zblob)
    prep_claims = np.concatenate([e[:c[0].prep_width] for e, c in zip(chip_evals, chips)])
    main_claims = np.concatenate([e[c[0].prep_width:] for e, c in zip(chip_evals, chips)])
    jblob, t_jag = timed(lambda: jp.prove_trusted_evaluations(z_row, [prep_claims, main_claims], [prep_data, sd], ch))
    print(json.dumps(dict(res, stages_ms={"commit": round(t_commit, 2), "logup_gkr": round(t_gkr, 2), "zerocheck": round(t_zc, 2),
                                          "jagged_evaluation_proof": round(t_jag, 2)},
                          proof_bytes={"logup_gkr": len(gblob), "zerocheck": len(zblob), "evaluation_proof": len(jblob)})), flush=True)


if __name__ == "__main__":
    main()
