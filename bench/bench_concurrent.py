#!/usr/bin/env python3
"""Shard-proving THROUGHPUT with several provers in flight on one GPU.

A shard proof is a chain of ~370 sumcheck rounds, each a small kernel plus a host round trip for the
transcript; about a third of its wall time the GPU idles on that latency. The library is re-entrant per
stream (stream-keyed arena, no global mutable state on the data path), so independent shards proved from
different host threads on different streams fill each other's gaps. This script proves the same synthetic
core-scale shard `--shards` times with 1 .. `--max-concurrency` worker threads and reports shards/s.

  python bench/bench_concurrent.py [--scale-log2 K] [--shards 8] [--max-concurrency 3]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bench"))

import torch  # noqa: E402

from sp1_amd import api  # noqa: E402
from synthetic_shard import build_shard  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale-log2", type=int, default=0)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--max-concurrency", type=int, default=3)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    L, lsh = 22 - args.scale_log2, 21 - args.scale_log2
    area_target = ((1 << 28) + (1 << 27)) >> (2 * args.scale_log2)
    chips, prep_prep, shapes, area = build_shard(L, lsh, area_target)
    jp = api.JaggedProver(L, lsh, 32, 2)
    prep_commit, prep_data = jp.commit_multilinears([prep_prep])
    torch.cuda.synchronize()

    def prove_one(stream):
        ch = api.DuplexChallenger()
        ch.observe(prep_commit)
        with torch.cuda.stream(stream):
            return api.prove_shard(chips, [], prep_data, L, lsh, 32, ch, stream=stream)

    ref = prove_one(torch.cuda.current_stream())          # warm-up + reference bytes
    torch.cuda.synchronize()
    for conc in range(1, args.max_concurrency + 1):
        streams = [torch.cuda.Stream() for _ in range(conc)]
        for st in streams:                                # the arena is keyed by stream: fill each one's cache first
            prove_one(st)
        todo = list(range(args.shards))
        lock = threading.Lock()
        bad = []

        def worker(s):
            while True:
                with lock:
                    if not todo:
                        return
                    todo.pop()
                if prove_one(s) != ref:
                    bad.append(1)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        threads = [threading.Thread(target=worker, args=(s,)) for s in streams]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert not bad, "a concurrently produced proof differs from the sequential one"
        print(json.dumps({"concurrency": conc, "shards": args.shards, "area_cells": area, "seconds": round(dt, 4),
                          "shards_per_s": round(args.shards / dt, 3), "ms_per_shard": round(1e3 * dt / args.shards, 2),
                          "cells_per_s": round(args.shards * area / dt)}), flush=True)


if __name__ == "__main__":
    main()
