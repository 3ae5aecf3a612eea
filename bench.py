#!/usr/bin/env python3
"""bench.py — core-shard commit hot path on MI355X, BASELINE.json config 2.

One "step" = BasefoldProver::commit_mles on one synthetic trace already resident in HBM
(column-major): Reed–Solomon encode (zero-padded NTT, log_blowup 2, bit-reversed) of a
2^20-row x 256-column KoalaBear trace + Poseidon2 Merkle commitment of the 2^22 x 256 codeword.
Metric: RISC-V cycles proved/sec — for the synthetic fixed-height trace one trace row stands for one
cycle (cycles only exist for real programs; the synthetic configs report rows/cells), so
value = rows committed per second summed over all GPUs. `config` also carries cells/s.

Usage: python bench.py --gpus N --steps K --warmup W     (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for `roofline` and `cpu_baseline`).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LG_N, WIDTH, LOG_BLOWUP, BATCH = 20, 256, 2, 32        # 8 stacked batches of 32 columns
HBM_PEAK_GBPS = 8000.0                                  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_TRAFFIC_PER_LAUNCH = 788688051                      # profiles/r01_pmc_summary.md: HBM bytes per leaf-hash launch (1/8 step)


def timers_read(api, name):
    n, ms = C.c_uint64(), C.c_double()
    api.check(api._L().sp1hip_timers_read(name.encode(), C.byref(n), C.byref(ms)))
    return n.value, ms.value


def effective_cores():
    """Host threads the container may actually use: min(cpu_count, affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_child(lg_rows):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    mles = [orc.random_felts((1 << lg_rows, BATCH), 42 + i) for i in range(WIDTH // BATCH)]
    orc.CommittedRound([m[:256] for m in mles], LOG_BLOWUP)          # warm-up / first touch
    t0 = time.perf_counter()
    orc.CommittedRound(mles, LOG_BLOWUP)
    print(json.dumps({"seconds": time.perf_counter() - t0}))


def cpu_baseline(lg_rows):
    """The CPU oracle (a port, NOT the reference binary) on a bounded sample of the same workload, in a
    child process so OpenMP is sized to the cores this container may use."""
    import subprocess
    cores = effective_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="false")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(lg_rows)], env=env,
                         capture_output=True, text=True, check=True).stdout
    dt = json.loads(out.strip().splitlines()[-1])["seconds"]
    return {"value": (1 << lg_rows) / dt, "unit": "cycles/s", "cores": cores, "kind": "port",
            "sample": "oracle commit_mles (RS encode + Poseidon2 Merkle) of 2^%d x %d rows = 1/%d of one step; "
                      "C++17 + OpenMP, %d threads (cgroup quota); %.2f s wall" % (lg_rows, WIDTH, 1 << (LG_N - lg_rows),
                                                                                 cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-sample-lg-rows", type=int, default=18)
    ap.add_argument("--cpu-baseline-child", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL; gloo lets "
                         "several ranks share one GPU when testing the N > 1 path on a 1-GPU box)")
    args = ap.parse_args()
    if args.cpu_baseline_child is not None:
        return cpu_baseline_child(args.cpu_baseline_child)

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    from sp1_amd import api
    L = api._L()

    # synthetic trace, resident in HBM, column-major: SplitMix64-style words < p, generated on device
    n = 1 << LG_N
    g = torch.Generator(device="cuda")
    g.manual_seed(42 + rank)
    mles = []
    for b in range(WIDTH // BATCH):
        words = torch.randint(0, api.P, (BATCH * n,), dtype=torch.int32, device="cuda", generator=g)
        mles.append(api.ColMajor(words, n, BATCH))
    prover = api.BasefoldProver(LOG_BLOWUP, 124, 16)

    def step():
        commit, pd = prover.commit_mles(mles)
        return commit, pd

    for _ in range(args.warmup):
        _, pd = step()
        del pd
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    api.check(L.sp1hip_timers_reset())
    api.check(L.sp1hip_timers_enable(1))
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last, pd = step()
        del pd
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    api.check(L.sp1hip_timers_enable(0))
    tl = {name: timers_read(api, name) for name in ("leaf_hash", "ntt_pass0", "ntt_pass1", "ntt_pass2", "compress")}
    from sp1_amd import shards
    dt = shards.max_over_ranks(dt)      # shards are striped one per rank: no data-path collective

    # the same kernels once more with the encode/hash overlap switched off (not timed into `value`): isolated
    # launch durations, so the roofline object can show both what a launch achieves alone and inside the step
    iso = {}
    if rank == 0 and world == 1:      # (N > 1 runs measure scaling only: no after-pass, no CPU baseline)
        os.environ["SP1HIP_COMMIT_OVERLAP"] = "0"
        api.check(L.sp1hip_timers_reset())
        api.check(L.sp1hip_timers_enable(1))
        t1 = time.perf_counter()
        iso_steps = max(2, min(5, args.steps))
        for _ in range(iso_steps):
            _, pd = step()
            del pd
        torch.cuda.synchronize()
        iso["ms_per_step"] = 1e3 * (time.perf_counter() - t1) / iso_steps
        api.check(L.sp1hip_timers_enable(0))
        for name in ("leaf_hash", "ntt_pass0", "ntt_pass1", "ntt_pass2"):
            k, m = timers_read(api, name)
            iso[name + "_ms_per_step"] = m / iso_steps
        del os.environ["SP1HIP_COMMIT_OVERLAP"]
        api.check(L.sp1hip_timers_reset())
    if world > 1:
        dist.barrier()

    if rank == 0:
        N = n << LOG_BLOWUP
        rows_per_s = world * args.steps * n / dt
        # dominant kernel: leaf hashing (one launch per stacked batch, 8 per step, overlapped with the encodes
        # of the following batches). Algorithmic bytes per step: leaf read 4*N*W + digest write 32*N;
        # per launch = 1/8 of that. The split adds 2 x 32 B of sponge-capacity carry per row per boundary.
        leaf_launches, leaf_ms_total = tl["leaf_hash"]
        per_step = leaf_launches // args.steps
        leaf_ms = leaf_ms_total / args.steps            # all leaf-hash launches of one step
        leaf_bytes = 4 * N * WIDTH + 32 * N
        carry_bytes = 64 * N * (per_step - 1)
        perms = N * (WIDTH // 8)
        ntt = {}
        for name in ("ntt_pass0", "ntt_pass1", "ntt_pass2", "compress"):
            k, m = tl[name]
            if k:
                ntt[name + "_ms_per_step"] = round(m / args.steps, 4)
        ntt_ms = sum(v for k, v in ntt.items() if k.startswith("ntt"))
        ntt_bytes = 4 * n * WIDTH * (1 + (1 << LOG_BLOWUP))
        valu_peak = 256 * 64 * 2.4e9
        iso_leaf = iso.get("leaf_hash_ms_per_step")
        iso_ntt = sum(iso[k] for k in iso if k.startswith("ntt"))
        out = {
            "metric": "RISC-V cycles proved/sec (core shard prove; synthetic config 2: commit phase, 1 trace row = 1 cycle)",
            "value": rows_per_s, "unit": "cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (KoalaBear Montgomery words, exact integer arithmetic)", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: synthetic 2^20-row x 256-col KoalaBear trace (8 stacked batches of 32), "
                                   "log_blowup 2: RS-encode NTT + Poseidon2 Merkle commit, bit-exact vs CPU oracle",
                       "rows": n, "cols": WIDTH, "log_blowup": LOG_BLOWUP, "parallelism": "independent shards, one per GPU",
                       "cells_per_s": rows_per_s * WIDTH, "commitment_word0": int(last[0])},
            "roofline": {"bound": "hbm", "kernel": "leaf_hash_part_kernel", "achieved": leaf_bytes / (leaf_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": leaf_bytes / (leaf_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         # HBM bytes per step of this kernel from the committed PMC passes
                         # (profiles/r01_pmc_summary.md: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
                         "traffic": PMC_TRAFFIC_PER_LAUNCH, "traffic_source": "profiles/r01_pmc_summary.md",
                         "launches_per_step": per_step, "avg_launch_ms": leaf_ms / max(per_step, 1),
                         "algorithmic_bytes_per_launch": leaf_bytes // max(per_step, 1),
                         "sponge_carry_bytes_per_step": carry_bytes,
                         # the bound that actually binds: VALU issue. 3573 = SQ_INSTS_VALU per permutation of a
                         # leaf_hash_part_kernel launch (profiles/r01_pmc_summary.md, last SQ pass; 3556 in the steady-state
                         # loop of the ISA listing + the carry load/store of the launch; the single-launch kernel of the
                         # isolated pass: 3558); peak = 256 CUs x 64 lanes x 2.4 GHz, one instruction per lane-clock.
                         # `frac` is measured inside the timed steps, where the launches share the chip with the
                         # encode passes of the following batches; `frac_isolated` is the same launch with the overlap
                         # switched off (SP1HIP_COMMIT_OVERLAP=0), measured right after the timed region.
                         "valu": {"insts_per_permutation": 3573, "achieved_lane_insts_per_s": 3573 * perms / (leaf_ms * 1e-3),
                                  "peak_lane_insts_per_s": valu_peak, "frac": 3573 * perms / (leaf_ms * 1e-3) / valu_peak,
                                  "frac_isolated": 3558 * perms / (iso_leaf * 1e-3) / valu_peak if iso_leaf else None},
                         "note": "integer-VALU-bound by construction (Poseidon2: %d permutations per step, "
                                 "%.3g permutations/s); see DESIGN.md" % (perms, perms / (leaf_ms * 1e-3)),
                         "rs_encode": {"bound": "hbm", "ms_per_step": ntt_ms, "algorithmic_bytes_per_step": ntt_bytes,
                                       "achieved": ntt_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None,
                                       "frac": ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if ntt_ms else None},
                         "per_step_ms": ntt,
                         "isolated": ({"note": "same workload, encode/hash overlap off, %d steps after the timed region" % iso_steps,
                                       "ms_per_step": round(iso["ms_per_step"], 4), "leaf_hash_ms_per_step": round(iso_leaf, 4),
                                       "rs_encode_ms_per_step": round(iso_ntt, 4)} if iso else None)},
        }
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_lg_rows) if (world == 1 and not args.no_cpu_baseline) else None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
