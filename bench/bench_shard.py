#!/usr/bin/env python3
"""Stage timings of the implemented hot path at CORE-shard scale (A = 2^28 + 2^27 cells,
stacking height 2^21, 32 columns per batch, log_blowup 2, 124 queries, 16 PoW bits).

Not the driver's bench (that is ../bench.py, BASELINE config 2). This script reports, for one
synthetic shard resident in HBM: jagged commit (stack + RS encode + Merkle), zerocheck over synthetic
chips covering the same area, and the BaseFold opening, so that the per-stage numbers in DESIGN.md /
profiles come from a measured run. LogUp-GKR and the jagged sumchecks are not implemented (§8f), so
this is NOT a complete shard proof.

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

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sp1_amd import api  # noqa: E402
from sp1_amd.air import AirProgram  # noqa: E402


def wide_air(width):
    """A degree-3 AIR over `width` columns: groups of 4 columns (a, b, c, d) with c = a*b and d*(d-1)*a = 0."""
    p = AirProgram("Wide%d" % width, width)
    for g in range(width // 4):
        a, b, c, d = (p.main(4 * g + k) for k in range(4))
        p.assert_eq(c, a * b)
        p.assert_zero(d * (d - 1) * a)
    return p


def wide_trace(rows, width, gen):
    """Satisfying trace, column-major on the device, Montgomery words (values are arbitrary field words;
    Montgomery mul of words == field mul of the represented values, so c = a*b is built with the library)."""
    cols = []
    L = api._L()
    for g in range(width // 4):
        a = torch.randint(0, api.P, (rows,), dtype=torch.int32, device="cuda", generator=gen)
        b = torch.randint(0, api.P, (rows,), dtype=torch.int32, device="cuda", generator=gen)
        # c = a * b in the field: compute on canonical values with int64 torch ops, then back to Montgomery
        ac, bc = a.clone(), b.clone()
        api.check(L.sp1hip_from_monty(api._dptr(ac), rows, api._stream_ptr()))
        api.check(L.sp1hip_from_monty(api._dptr(bc), rows, api._stream_ptr()))
        prod = ((ac.to(torch.int64) * bc.to(torch.int64)) % api.P).to(torch.int32)
        api.check(L.sp1hip_to_monty(api._dptr(prod), rows, api._stream_ptr()))
        d = torch.randint(0, 2, (rows,), dtype=torch.int32, device="cuda", generator=gen)
        api.check(L.sp1hip_to_monty(api._dptr(d), rows, api._stream_ptr()))
        cols += [a, b, prod, d]
    return api.ColMajor(torch.cat(cols), rows, width)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale-log2", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    L = 22 - args.scale_log2            # max_log_row_count
    lsh = 21 - args.scale_log2          # log stacking height
    area_target = ((1 << 28) + (1 << 27)) >> (2 * args.scale_log2) if args.scale_log2 else (1 << 28) + (1 << 27)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    # chips: widths like real ones (narrow & tall ... wide & short), heights multiples of 32, <= 2^L
    shapes, area = [], 0
    widths = [8, 16, 32, 64, 100, 200, 400]
    k = 0
    while area < area_target:                      # power-of-two heights (the column-eval kernel needs them)
        w = widths[k % len(widths)] // 4 * 4
        rows = (1 << L) >> (k % 4)
        while rows * w > area_target - area and rows > 32:
            rows >>= 1
        shapes.append((rows, w))
        area += rows * w
        k += 1
    tables = [wide_trace(r, w, gen) for r, w in shapes]
    airs = [wide_air(w) for _, w in shapes]
    print("chips:", shapes, "area = %.3e cells" % area, file=sys.stderr)

    res = {"area_cells": area, "max_log_row_count": L, "log_stacking_height": lsh, "chips": len(shapes)}
    jp = api.JaggedProver(L, lsh, 32, 2)
    prover = api.BasefoldProver(2, 124, 16)
    for rep in range(args.repeat):
        (commit, sd), t_commit = timed(lambda: jp.commit_multilinears(tables))
        ch = api.DuplexChallenger()
        ch.observe(commit)
        zeta = ch.sample_point(L)
        alpha, gkr = ch.sample_ext_element(), ch.sample_ext_element()
        # trace-column evaluations at zeta (what LogUp-GKR hands to zerocheck): eval of zero-padded columns
        def openings():
            eq = api.device_words(4 << L)
            api.check(api._L().sp1hip_partial_lagrange(api._ext_array(zeta), L, api._dptr(eq), api._stream_ptr()))
            outs = []
            for t in tables:
                # columns are shorter than 2^L: evaluate against the first `rows` entries of eq == zero padding
                o = api.device_words(t.width * 4)
                # build an eq table truncated to t.height (SoA): 4 slices
                eq_t = torch.cat([eq[kk << L:(kk << L) + t.height] for kk in range(4)])
                lg = t.height.bit_length() - 1
                assert 1 << lg == t.height
                api.check(api._L().sp1hip_mle_eval_columns(api._tensor_array([t]), 1, lg, api._dptr(eq_t), api._dptr(o),
                                                           api._stream_ptr()))
                outs.append(api.to_host(o, (t.width, 4)))
            return outs
        ops, t_open_evals = timed(openings)
        chips = [api.ZerocheckChip(a, t) for a, t in zip(airs, tables)]
        blob, t_zc = timed(lambda: api.zerocheck(chips, L, zeta, np.concatenate(ops), alpha, gkr, [], ch))
        # jagged PCS evaluation proof at the zerocheck point: jagged sumcheck + jagged-eval sumcheck + stacked
        # batch evaluations + BaseFold opening (= ShardProof.evaluation_proof)
        z_row, chip_evals = api.parse_zerocheck_proof(blob)
        main_claims = np.concatenate(chip_evals)         # no preprocessed columns here
        api.check(api._L().sp1hip_timers_enable(1))
        api.check(api._L().sp1hip_timers_reset())
        proof, t_jag = timed(lambda: jp.prove_trusted_evaluations(z_row, [main_claims], [sd], ch))
        stages = {}
        for name in ("jagged_round0_sum", "jagged_fold0_sum", "jagged_fold_sum", "jagged_batch_evals"):
            cnt, ms = C.c_uint64(), C.c_double()
            api.check(api._L().sp1hip_timers_read(name.encode(), C.byref(cnt), C.byref(ms)))
            stages[name + "_ms"] = round(ms.value, 3)
        api.check(api._L().sp1hip_timers_enable(0))
        res = dict(res, commit_ms=t_commit, zerocheck_ms=t_zc, jagged_eval_proof_ms=t_jag, jagged_kernels=stages,
                   zerocheck_proof_bytes=len(blob), jagged_proof_bytes=len(proof), trace_openings_ms=t_open_evals)
        print(json.dumps(res), flush=True)
        del sd


if __name__ == "__main__":
    main()
