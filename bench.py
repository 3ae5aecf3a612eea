#!/usr/bin/env python3
"""bench.py — one WHOLE core-shard proof per step on MI355X (BASELINE.json metric: core shard prove).

One "step" = `sp1hip_prove_shard` = `ShardProver::prove_shard_with_data`
(/root/reference/crates/hypercube/src/prover/shard.rs:L650-L792) on one synthetic core shard already resident in HBM
(column-major traces): jagged commit of the main traces (RS-encode NTT, blowup 4 + Poseidon2 Merkle) -> LogUp-GKR ->
zerocheck -> jagged evaluation proof (jagged sumcheck, jagged-eval, stacked BaseFold opening, 124 queries, 16-bit PoW);
the output is a complete bincode(ShardProof) that the pinned verifier accepts (tests/test_gpu_shard.py).

Workload (`config.workload`, default since round 5): a FULL CORE SHARD OF THE REFERENCE'S `fibonacci` GUEST — the program
BASELINE.json's metric is quoted on — executed by the rv64im executor of libsp1hip.so (sp1_amd/csrc/rv64_exec.cpp) and traced
by sp1_amd/machines/riscv_exec.py: 2^23 executed cycles (the power of two below the 8.8e6 cycles at which the reference's area
threshold of 2^28 + 2^27 cells cuts a fibonacci shard: bench/program_shard.py), 4.1e8 trace cells, every chip a real chip of
the RISC-V machine with the constraints and interactions transcribed from the reference's `Air::eval` bodies, the shard's own
public values. `value` = RISC-V CYCLES PROVED PER SECOND, the reference's "Core kHz" x 1000
(/root/reference/sp1-gpu/crates/perf/src/report.rs:L52-L60: cycles / core-proving seconds); `cells_per_s` rides along.
`--workload loop|keccak` are the other guests of the reference's perf harness this executor runs. The earlier workloads stay:
`--workload real` (rounds 4-5: the RISC-V chips at the heights of the reference's recorded core shard 0, bench/core_real.py, on
traces of a synthetic rv64im loop executed by riscv_trace.py; `value` = cells/s), `precompile` (a Keccak precompile shard),
`core` (rounds 1-3: synthetic constraints on the recorded widths).

Usage: python bench.py --gpus N --steps K --warmup W     (N > 1: launched by torch.distributed.run, shards striped
one per rank, no data-path collective). Prints ONE JSON line on rank 0; see DESIGN.md §8 for every field.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bench"))

CORE_AREA = (1 << 28) + (1 << 27)
HBM_PEAK_GBPS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_GUIDE = 256 * 4 * 32 * 2.4e9   # guide: 4 SIMD-32 per CU, a wave64 instruction issues in 2 cycles -> 78.6 T lane-inst/s
VALU_PEAK_MEASURED = 256 * 64 * 2.4e9    # measured VOP3-integer / f64 rate (profiles/r01_ubench*.txt): one lane-inst per lane-clock
TIMERS = ("ntt_pass0", "ntt_pass1", "ntt_pass2", "leaf_hash", "compress", "gkr_first_layer", "gkr_transition", "gkr_pass_sum", "gkr_pass_fold_sum", "gkr_pass_fold",
          "gkr_openings", "zerocheck_round", "zerocheck_fix", "jagged_round0_sum", "jagged_fold0_sum",
          "jagged_fold_sum", "jagged_batch_evals", "stage_commit", "stage_logup_gkr", "stage_zerocheck", "stage_evaluation_proof")


def thread_cpu_times():
    """{tid: (user + system seconds, comm)} of this process's threads (/proc/self/task)."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{tid}/stat") as f:
                    st = f.read()
                comm = st[st.index("(") + 1:st.rindex(")")]
                fields = st[st.rindex(")") + 2:].split()
                out[int(tid)] = ((int(fields[11]) + int(fields[12])) / tick, comm)
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return out


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


def programs_of(kind, names):
    """(AirProgram, InteractionProgram) list of a workload, rebuilt by name in a child process that has no traces."""
    if kind in ("real", "precompile") + PROGRAMS:
        import core_real
        return core_real.programs_for(names)
    if kind == "core":
        from core_shard import chip_programs, load_shape
        return [chip_programs(c["name"], c["width"], c["prep_width"], c["constraints"], c["interactions"])
                for c in sorted(load_shape()["chips"], key=lambda c: c["name"])]
    from sp1_amd.machines import recursion as R
    return R.compress_machine()


PROGRAMS = ("fibonacci", "loop", "keccak", "sha2", "poseidon2", "rsp")   # real guest programs: bench/program_shard.py


RISCV_KINDS = ("real", "precompile") + PROGRAMS     # workloads of the rv64im machine: their shards carry PublicValues (meta["publics"])


def publics_of(kind):
    """The synthetic shards carry no public values; an rv64im shard's are its own (meta["publics"]: the record-level statement
    `eval_public_values` closes the shard's buses with — sp1_amd/machines/public_values.py)."""
    import numpy as np
    assert kind not in RISCV_KINDS, "an rv64im shard is proved with ITS public values"
    return np.zeros(0, np.uint32)


def pv_program_of(kind):
    """The machine's eval_public_values for the verifier: the RISC-V machine has one, the recursion machine and the synthetic shards none."""
    if kind not in RISCV_KINDS:
        return None
    from sp1_amd.machines import public_values as PVM
    return PVM.verifier_program()


def build_workload(kind, k, L, seed):
    """chips [(air, inter, main ColMajor, prep ColMajor | None)], meta — `kind` "real" (the rv64im core shard), "precompile" (a Keccak
    precompile shard: the wide-chip regime) or "core" (synthetic)."""
    if kind == "real":
        import core_real
        return core_real.build_real_shard(scale=1.0 / (1 << (2 * k)), seed=seed)
    if kind == "precompile":
        import precompile_shard
        return precompile_shard.build_precompile_shard(max(1, precompile_shard.FULL_EVENTS >> (2 * k)), seed=seed)
    if kind in PROGRAMS:
        import program_shard
        return program_shard.build_program_shard(kind, k, shard_index=seed - 42)     # rank r proves shard r of the same execution
    from core_shard import build_core_shard
    return build_core_shard(CORE_AREA >> (2 * k), L, seed=seed)


# ---- CPU baseline: the oracle's prove_shard_with_data on a scaled-down copy of the same shard -------------------------
def cpu_baseline_child(path):
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    z = np.load(path)
    L, lsh = int(z["L"]), int(z["lsh"])
    chips = [(air, inter, z["main%d" % k], z["prep%d" % k] if air.prep_width else None)
             for k, (air, inter) in enumerate(programs_of(str(z["kind"]), [str(n) for n in z["names"]]))]
    t0 = time.perf_counter()
    prep = orc.JaggedRound([c[3] for c in chips if c[3] is not None], L, lsh, 32, 2)
    t_setup = time.perf_counter() - t0
    ch = orc.Challenger()
    ch.observe(prep.commit)
    # LogUp-GKR over the chips' REAL rows with the padding in closed form — the shape of the reference's CPU prover
    # (crates/hypercube/src/logup_gkr/execution.rs:L112-L382); the oracle's dense formulation (its independent check of the
    # GPU algorithm) costs 2^L / rows more and is not what a CPU prover does. ONE pass (no size query).
    orc.set_gkr_sparse(True)
    t0 = time.perf_counter()
    publics = z["publics"] if "publics" in z.files else publics_of(str(z["kind"]))
    blob = orc.shard_prove(chips, publics, prep, L, lsh, 32, ch, 2, 124, 16, capacity=64 << 20)
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "setup_seconds": t_setup, "proof_bytes": len(blob), "stage_seconds": orc.stage_seconds()}))


def cpu_sample(api, scale_log2, cores, kind="real"):
    import subprocess
    import tempfile

    import numpy as np
    L, lsh = max(22 - scale_log2, 17), 21 - scale_log2
    chips, meta = build_workload(kind, scale_log2, L, 42)
    arrays = {"L": L, "lsh": lsh, "kind": kind, "names": np.array([c[0].name for c in chips])}
    if "publics" in meta:
        arrays["publics"] = meta["publics"]
    for k, (_, _, main, prep) in enumerate(chips):
        arrays["main%d" % k] = main.to_row_major_host()
        if prep is not None:
            arrays["prep%d" % k] = prep.to_row_major_host()
    del chips
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "shard.npz")
        np.savez(path, **arrays)
        del arrays
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="false")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", path], env=env,
                             capture_output=True, text=True, check=True).stdout
    r = json.loads(out.strip().splitlines()[-1])
    r["cells"] = meta["area_cells"]
    r["cells_per_s"] = meta["area_cells"] / r["seconds"]
    r["cycles"] = meta.get("cycles")
    r["max_log_row_count"] = L
    return r


def cpu_baseline(api, scale_log2, kind="real"):
    """The CPU oracle (a C++17/OpenMP restatement of the reference's prover, NOT the reference binary) proving the same
    shard shape scaled down by 4^scale_log2 — at a scale where its time grows with the area: the rate at a quarter of the
    sample rides along as the check (cells/s within 2x between 1/256 and 1/64) — in a child process so OpenMP is
    sized to the cores this container may use; per-stage seconds from the oracle's own timers."""
    cores = effective_cores()
    big = cpu_sample(api, scale_log2, cores, kind)
    small = cpu_sample(api, scale_log2 + 1, cores, kind)
    # the same unit as `value`: cycles/s for a guest program's shard, cells/s otherwise
    head = ({"value": big["cycles"] / big["seconds"], "unit": "cycles/s", "cycles": big["cycles"], "cells_per_s": big["cells_per_s"]}
            if kind in PROGRAMS else {"value": big["cells_per_s"], "unit": "cells/s"})
    return {**head, "cores": cores, "kind": "port",
            "stage_seconds": {k: round(v, 3) for k, v in big["stage_seconds"].items()},
            "seconds": round(big["seconds"], 2), "cells": big["cells"], "sample_fraction": "1/%d" % (1 << (2 * scale_log2)),
            "simd": "AVX-512 leaf hash / Merkle compress (16 permutations per register) and RS-encode butterflies (16 columns) in the oracle's own "
                    "kb_simd.hpp where the CPU has them; the sumchecks are scalar C++ + OpenMP",
            "quarter_sample": {"sample_fraction": "1/%d" % (1 << (2 * scale_log2 + 2)), "cells": small["cells"], "seconds": round(small["seconds"], 2),
                               "cells_per_s": small["cells_per_s"], "ratio_to_value": small["cells_per_s"] / big["cells_per_s"]},
            "sample": "oracle prove_shard_with_data (commit + LogUp-GKR over real rows + zerocheck + jagged evaluation proof, 124 queries, "
                      "16-bit PoW; one pass) of the same real-chip shard (same machine, heights scaled) at 1/%d of the area (%d cells, max_log_row_count %d); "
                      "C++17 + OpenMP, %d threads (cgroup quota); %.2f s wall"
                      % (1 << (2 * scale_log2), big["cells"], big["max_log_row_count"], cores, big["seconds"])}


# ---- the pinned verifier on the timed proof (checker only, untimed, in a child process) --------------------------------
def verify_child(path):
    """oracle shard_verify (the restated ShardVerifier::verify_shard that accepts the reference's real ShardProof) on one
    proof: every Merkle opening, fold, sumcheck round, lookup balance and constraint evaluation. Prints {"rc": 0} if accepted."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    z = np.load(path)
    L, lsh, kind = int(z["L"]), int(z["lsh"]), str(z["kind"])
    progs = programs_of(kind, [str(n) for n in z["names"]])
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
              for a, i in progs]
    ch = orc.Challenger()
    ch.observe(z["commit"])
    if "vk_words" in z.files and z["vk_words"].size:         # the whole verifying key (vk.observe_into): + pc_start, digest, flag, padding
        ch.observe(np.concatenate([z["vk_words"], np.zeros(7, np.uint32)]))
    t0 = time.perf_counter()
    rc = orc.shard_verify(shapes, z["commit"], z["proof"].tobytes(), L, lsh, ch, 2, 124, 16, pv_program=pv_program_of(kind))
    print(json.dumps({"rc": int(rc), "seconds": time.perf_counter() - t0,
                      "state_matches": bool(z["state"].size == 0 or np.array_equal(ch.state(), z["state"]))}))


def verify_proof(kind, proof, commit, state, L, lsh, names=(), vk_words=None):
    """Runs verify_child on `proof`; returns True iff the verifier accepts AND ends in the prover's transcript state. `vk_words`: the
    rest of the verifying key the transcript absorbed behind the commitment (vk.observe_into: pc_start, the initial global
    cumulative sum; the flag and the padding are zero) — None when only the commitment was observed; `state` may be None."""
    import subprocess
    import tempfile

    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "proof.npz")
        np.savez(path, proof=np.frombuffer(proof, np.uint8), commit=np.asarray(commit, np.uint32), state=np.asarray(state if state is not None else [], np.uint32),
                 L=L, lsh=lsh, kind=kind, names=np.array(list(names), dtype=str), vk_words=np.asarray(vk_words if vk_words is not None else [], np.uint32))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--verify-child", path], capture_output=True, text=True)
    if r.returncode != 0:
        print("verifier child failed:\n" + r.stderr[-2000:], file=sys.stderr)
        return False
    v = json.loads(r.stdout.strip().splitlines()[-1])
    return v["rc"] == 0 and v["state_matches"]


def real_machine(api, repeat=4):
    """The one REAL machine in the tree at its real shape (bench/bench_recursion.py): the eight chips of
    RecursionAir::compress_machine() with the table heights of the reference's own compress proof (8.9e7 cells), real
    constraints and interactions, satisfying traces. Whole sp1hip_prove_shard, traces resident in HBM."""
    import torch
    from sp1_amd.machines import recursion as R, recursion_trace as RT
    L, lsh = 21, 20
    tabs, pv = RT.generate(dict(RT.REFERENCE_COMPRESS_HEIGHTS), seed=1)
    m = R.compress_machine()
    area = int(sum(p.size + mm.size for p, mm in tabs.values()))
    dev = [(a, i, api.ColMajor.from_row_major_host(tabs[a.name][1]), api.ColMajor.from_row_major_host(tabs[a.name][0])) for a, i in m]
    del tabs
    jp = api.JaggedProver(L, lsh, 32, 2)
    commit, prep = jp.commit_multilinears([d[3] for d in dev])
    times, proof, ch = [], None, None
    for _ in range(repeat + 1):
        ch = api.DuplexChallenger()
        ch.observe(commit)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proof = api.prove_shard(dev, pv, prep, L, lsh, 32, ch)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    ms = 1e3 * sum(times[1:]) / repeat                      # the first proof warms the arenas
    return {"machine": "recursion compress machine (8 chips, 220 constraints, 51 interactions; sp1_amd/machines/recursion.py, pinned "
                       "by the reference's own ShardProof), table heights of the reference's compress proof",
            "cells": area, "ms_per_proof": ms, "cells_per_s": area / (ms * 1e-3), "proof_bytes": len(proof), "proofs_timed": repeat,
            "verified": verify_proof("recursion", proof, commit, ch.state(), L, lsh)}


def synthetic_core_shaped(api, k, repeat=3):
    """The round-1..3 workload (bench/core_shard.py: synthetic constraints on the widths / counts of RISC-V chips) for continuity."""
    import torch
    from core_shard import build_core_shard
    L, lsh = 22 - k, 21 - k
    chips, meta = build_core_shard(CORE_AREA >> (2 * k), L, seed=42)
    jp = api.JaggedProver(L, lsh, 32, 2)
    commit, prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
    times = []
    for _ in range(repeat + 1):
        ch = api.DuplexChallenger()
        ch.observe(commit)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        api.prove_shard(chips, [], prep, L, lsh, 32, ch)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    ms = 1e3 * sum(times[1:]) / repeat
    return {"workload": "core-shaped synthetic shard of rounds 1-3 (33 chips, 730 interactions, 1605 synthetic constraints)",
            "cells": meta["area_cells"], "ms_per_proof": ms, "cells_per_s": meta["area_cells"] / (ms * 1e-3)}


def tree_mode(api, shards, dist, torch, args, rank, world, use_dist, chips, meta, prep_commit, prep_data, L, lsh, publics, kind, names):
    """BASELINE config 5's shape (`CompressTree::reduce_proofs`, crates/prover/src/worker/controller/compress.rs:L234-L470) on the
    backends this box has: K core shards pulled from a work queue by the ranks (each rank proves ITS shard's traces, with a different
    transcript head per leaf so that the leaf proofs differ), the proofs reduced to one root as they arrive — every parent a
    ShardProof of the reference's recursion compress machine that commits to its children (shards.recursion_combine), proven by
    whichever rank takes the reduce task with sp1hip_prove_shard; children travel point to point (RCCL send / recv under nccl, gloo
    otherwise). The WHOLE job is inside the timed region; what every rank did is collected at the end."""
    import numpy as np
    from sp1_amd.machines import recursion as R
    from sp1_amd.machines import recursion_trace as RT
    k = args.scale_log2
    n_leaves = args.tree_leaves or 2 * world + 1
    # the recursion machine of a tree node: the reference's compress shape scaled like the leaves
    counts = {n: max(8, h >> (2 * k)) for n, h in RT.REFERENCE_COMPRESS_HEIGHTS.items() if n != "PublicValues"}
    rL, rlsh = max(21 - k, 8), max(20 - k, 7)
    rmachine = R.compress_machine()
    node_jp = api.JaggedProver(rL, rlsh, 32, 2)

    def prove_node(tables, pv):
        # a node's recursion program (its preprocessed tables carry the digest rows) is set up INSIDE the timed region, like the
        # reference's per-node program; the blob that travels = the node's preprocessed commitment + bincode(ShardProof)
        dev = [(a, i, api.ColMajor.from_row_major_host(tables[a.name][1]), api.ColMajor.from_row_major_host(tables[a.name][0])) for a, i in rmachine]
        commit, prep = node_jp.commit_multilinears([d[3] for d in dev])
        ch = api.DuplexChallenger()
        ch.observe(commit)
        return bytes(np.asarray(commit, np.uint32).tobytes()) + api.prove_shard(dev, pv, prep, rL, rlsh, 32, ch)
    combine = shards.recursion_combine(prove_node, counts, seed=7)

    def leaf(i):
        ch = api.DuplexChallenger()
        ch.observe(prep_commit)
        ch.observe(np.array([i + 1], dtype=np.uint32))      # a per-leaf transcript head: K different proofs of the rank's shard
        return api.prove_shard(chips, publics, prep_data, L, lsh, 32, ch)
    # warm-up: one leaf and one node (arena, plans, pinned slots)
    leaf(10 ** 6)
    combine([b"warm", b"up"])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    # the reference's shape (sp1_amd/scheduler.py): ranks PULL leaves from a queue, a finished proof joins the adjacent range
    # that is waiting for it, a full group becomes a reduce task on the same queue; no static assignment, no barrier per level
    from sp1_amd import scheduler
    wq = scheduler.WorkQueue(n_leaves, 2, name="bench-tree")
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    root, st = wq.run(leaf, combine)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = shards.max_over_ranks(time.perf_counter() - t0)
    per_rank = {"leaves": len(st["leaves"]), "joins": len(st["joins"]), "busy_ms": 1e3 * st["busy_s"], "wait_ms": 1e3 * st["wait_s"],
                "sent_bytes": st["sent_bytes"], "recv_bytes": st["recv_bytes"], "store_ops": st["store_ops"]}
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank)
    else:
        gathered = [per_rank]
    if rank != 0:
        return None
    scheduler.check_tree(n_leaves, st["nodes"], 2)
    # the root is a ShardProof of the recursion machine: full verification by the oracle (untimed)
    verified = None
    if not args.no_verify and root is not None:
        verified = verify_tree_root(root, rL, rlsh)
    area = meta["area_cells"]
    depth = {i: 0 for i in range(n_leaves)}
    for nd in st["nodes"]:
        depth[nd["pid"]] = 1 + max(depth[c] for c in nd["children"])
    return {"metric": "core shards proved AND reduced to one root proof through the recursion compress tree: leaf trace cells / s (whole job)",
            "value": n_leaves * area / dt, "unit": "cells/s", "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": 1e3 * dt,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 (KoalaBear Montgomery words)",
            "data": "synthetic", "mode": "tree",
            "config": {"workload": "%d leaves (%s shard, scale 4^-%d, %d cells each) pulled from a work queue by %d ranks + compress tree joined as "
                                   "proofs arrive (adjacent ranges, arity 2): every node a ShardProof of the recursion machine (%d cells) that commits to "
                                   "its children's bytes — a STAND-IN program: the recursion verifier circuit is Rust and out of scope"
                                   % (n_leaves, kind, k, area, world, sum(counts.values())),
                       "backend": args.backend if use_dist else None, "leaves": n_leaves, "arity": 2, "scheduler": "work queue + event-driven tree (no level barrier)"},
            "tree": {"nodes": len(st["nodes"]), "depth": max(depth.values()), "per_rank": gathered, "root_bytes": len(root) if root else None,
                     "root_verified": verified}}


def verify_tree_root(root, rL, rlsh):
    import subprocess
    import tempfile

    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "root.npz")
        commit = np.frombuffer(root[:32], dtype=np.uint32).copy()
        ch_state = np.zeros(0, np.uint32)
        np.savez(path, proof=np.frombuffer(root[32:], np.uint8), commit=commit, state=ch_state, L=rL, lsh=rlsh, kind="recursion",
                 names=np.array([], dtype=str))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--verify-child", path], capture_output=True, text=True)
    if r.returncode != 0:
        print("verifier child failed:\n" + r.stderr[-2000:], file=sys.stderr)
        return False
    return json.loads(r.stdout.strip().splitlines()[-1])["rc"] == 0


class GpuSampler:
    """Clock / power of GPU 0 from sysfs while a phase runs (amdgpu hwmon: power1_average or power1_input in microwatts,
    freq1_input = sclk in Hz): whether several provers in flight run into the package's power management
     is answered by these samples next to the per-proof times, not by a guess."""

    def __init__(self, period=0.05):
        import glob
        self.period, self.samples, self._stop, self._thread = period, [], threading.Event(), None
        # the box has several GPUs in sysfs and one visible to HIP: pick the hwmon of OUR device by its PCI address
        base = "/sys/class/drm/card*/device"
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            if os.path.isdir("/sys/bus/pci/devices/" + addr):
                base = "/sys/bus/pci/devices/" + addr
        except Exception:
            addr = None
        self.device = addr
        self.power = next(iter(sorted(glob.glob(base + "/hwmon/hwmon*/power1_average") + glob.glob(base + "/hwmon/hwmon*/power1_input"))), None)
        self.freq = next(iter(sorted(glob.glob(base + "/hwmon/hwmon*/freq1_input"))), None)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def __enter__(self):
        def run():
            while not self._stop.is_set():
                self.samples.append((self._read(self.power), self._read(self.freq)))
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join()

    def summary(self):
        def stats(xs, scale):
            xs = [x * scale for x in xs if x is not None]
            return {"mean": sum(xs) / len(xs), "min": min(xs), "max": max(xs)} if xs else None
        return {"pci": self.device, "samples": len(self.samples), "power_w": stats([p for p, _ in self.samples], 1e-6),
                "sclk_mhz": stats([f for _, f in self.samples], 1e-6)}


def in_flight(api, chips, area, L, lsh, n_proofs, publics=(), alu_events=None):
    """The library's prover pool with 1 to 4 slots on the SAME resident shard: throughput with N proofs in flight, the
    per-proof proving times in completion order, and the GPU's clock / power while each phase runs. Proofs are checked
    against `sp1hip_prove_shard_with_pk` called directly."""
    import torch
    prep_tables = [c[3] for c in chips if c[3] is not None]
    pk = api.ProvingKey(prep_tables, L, lsh, 32)
    want = pk.prove_shard(chips, publics)
    torch.cuda.synchronize()
    out = {"proofs_per_phase": n_proofs, "slots": {}}
    for n in (1, 2, 3, 4):
        released = C.c_size_t()
        api.check(api._L().sp1hip_mem_trim(C.byref(released)))       # every phase starts from an empty arena and refills it in its warm-up
        pool = api.ProverPool(n)
        for t in [pool.submit(pk, chips, publics) for _ in range(n)]:          # fill every slot's arena
            assert pool.wait(t)[0] == want, "a pool proof differs from the direct one"
        torch.cuda.synchronize()
        with GpuSampler() as smp:
            t0 = time.perf_counter()
            tickets = [pool.submit(pk, chips, publics) for _ in range(n_proofs)]
            res = [pool.wait(t) for t in tickets]
            dt = time.perf_counter() - t0
        pool.close()
        assert all(r[0] == want for r in res), "a pool proof differs from the direct one"
        out["slots"][str(n)] = {"ms_per_proof": 1e3 * dt / n_proofs, "proofs_per_s": n_proofs / dt, "cells_per_s": n_proofs * area / dt,
                                "proving_ms_each": [round(r[1]["proving_ms"], 1) for r in res], "gpu": smp.summary()}
    # PCIe-inclusive: the same pool (3 slots) fed with the main traces as PINNED HOST row-major tables — what a host trace
    # generator hands over; the stager thread uploads and transposes shard k+1.. while the slots prove (sp1hip_stage_tables)
    host = [(a, i, api.PinnedHost(m.to_row_major_host()) if m is not None else None, pr) for (a, i, m, pr) in chips]
    host_bytes = sum(4 * c[2].shape[0] * c[2].shape[1] for c in host if c[2] is not None)
    pool = api.ProverPool(3)
    for t in [pool.submit(pk, host, publics) for _ in range(3)]:
        assert pool.wait(t)[0] == want, "a staged pool proof differs from the direct one"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = [pool.wait(t) for t in [pool.submit(pk, host, publics) for _ in range(n_proofs)]]
    dt = time.perf_counter() - t0
    pool.close()
    assert all(r[0] == want for r in res), "a staged pool proof differs from the direct one"
    out["staged_from_host"] = {"slots": 3, "ms_per_proof": 1e3 * dt / n_proofs, "cells_per_s": n_proofs * area / dt,
                               "host_trace_bytes_per_proof": host_bytes, "staging_ms_each": [round(r[1]["staging_ms"], 1) for r in res],
                               "note": "pinned row-major host traces -> column-major device tables inside the pipeline (PCIe-inclusive; never `value`)"}
    # The instruction chips generated ON THE DEVICE from event records (sp1hip_tracegen_riscv_alu, round 6): per proof the host hands
    # over 88-byte events for Add / Addi / Sub / Addw / Subw / Mul / ShiftRight / Branch — 99.6 % of a fibonacci shard's rows — and
    # pinned tables only for the other chips; the upload and the tracegen of proof k + 1 run on the caller's stream while the
    # pool's slots prove proofs k, k - 1, k - 2. Tables must equal the host-made ones word for word, proofs the direct one.
    if alu_events:
        try:
            import torch as _t
            pinned = {n: _t.from_numpy(ev).pin_memory() for n, ev in alu_events.items()}
            height = {c[0].name: c[2].height for c in chips}
            resident = {c[0].name: c[2] for c in chips}

            def make_tables():
                tabs = {n: api.tracegen_riscv_alu(n, p_.cuda(non_blocking=True), height[n]) for n, p_ in pinned.items()}
                _t.cuda.current_stream().synchronize()                   # the pool proves on its own streams
                return tabs
            tabs0 = make_tables()
            same = all(bool(_t.equal(tabs0[n].words, resident[n].words)) for n in tabs0)
            tms = []
            for _ in range(3):
                _t.cuda.synchronize()
                t1 = time.perf_counter()
                make_tables()
                tms.append(1e3 * (time.perf_counter() - t1))
            host3 = [(a, i, None if a.name in pinned else m, pr) for (a, i, m, pr) in host]
            ev_bytes = sum(int(p_.numel()) * 8 for p_ in pinned.values())
            rest_bytes = sum(4 * c[2].shape[0] * c[2].shape[1] for c in host3 if c[2] is not None)
            pool = api.ProverPool(3)
            submit = lambda: pool.submit(pk, [(a, i, tabs_[a.name] if a.name in pinned else m, pr) for (a, i, m, pr) in host3], publics)
            for _ in range(3):                                           # warm-up: every slot's arena
                tabs_ = make_tables()
                assert pool.wait(submit())[0] == want, "a proof from device-generated instruction tables differs"
            _t.cuda.synchronize()
            t0 = time.perf_counter()
            tickets, res, keep = [], [], []
            for _ in range(n_proofs):
                tabs_ = make_tables()
                keep.append(tabs_)
                tickets.append(submit())
                if len(tickets) > 3:                                     # at most three in flight, like the slots
                    res.append(pool.wait(tickets.pop(0)))
                    keep.pop(0)
            res += [pool.wait(t) for t in tickets]
            dt = time.perf_counter() - t0
            pool.close()
            assert all(r[0] == want for r in res), "a proof from device-generated instruction tables differs"
            out["staged_events"] = {
                "slots": 3, "ms_per_proof": 1e3 * dt / n_proofs, "cells_per_s": n_proofs * area / dt, "chips_on_device": sorted(pinned),
                "event_bytes_per_proof": ev_bytes, "host_trace_bytes_per_proof": rest_bytes, "tables_replaced_bytes": host_bytes - rest_bytes,
                "upload_and_tracegen_ms": min(tms), "device_tables_equal_host_traces": same,
                "rows_on_device": sum(int(p_.shape[0]) for p_ in pinned.values()),
                "note": "88-byte event records over PCIe + device trace generation for the instruction chips, pinned host tables for the rest "
                        "(PCIe-inclusive; never `value`); compare staged_from_host (every table over PCIe) and slots['3'] (everything resident)"}
            del keep, tabs0, pinned
        except Exception as e:                        # an extra: never the reason a line is missing
            print("bench.py: staged_events not measured: %r" % (e,), file=sys.stderr)
    # The Global chip generated ON THE DEVICE (sp1hip_tracegen_riscv_global: a third of a core shard's cells never cross PCIe):
    # events are read back from the resident table once (setup), the device table must equal it word for word, then the same
    # staged pipeline runs with every OTHER trace coming from pinned host memory
    gi = next((k for k, c in enumerate(chips) if c[0].name == "Global" and c[0].main_width == 241), None)
    if gi is not None:
        try:
            import numpy as np
            g = chips[gi][2]
            gw = g.words.view(g.width, g.height)
            lay = chips[gi][0].layout
            r_inv = pow(1 << 32, -1, api.P)
            canon = lambda col: ((gw[col].to(torch.int64) & 0xFFFFFFFF) * r_inv) % api.P
            n_ev = int(canon(lay["is_real"]).sum())
            ev = torch.stack([canon(lay["message"] + j)[:n_ev] for j in range(8)] + [(canon(lay["is_receive"]) | (canon(lay["kind"]) << 8))[:n_ev]], dim=1)
            ev = ev.to(torch.int32).contiguous()
            tms = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                table = api.tracegen_riscv_global(ev, g.height)
                torch.cuda.synchronize()
                tms.append(1e3 * (time.perf_counter() - t1))
            same = bool(torch.equal(table.words, g.words))
            host2 = [(a, i, (table if k == gi else m), pr) for k, (a, i, m, pr) in enumerate(host)]
            pool = api.ProverPool(3)
            for t in [pool.submit(pk, host2, publics) for _ in range(3)]:
                assert pool.wait(t)[0] == want, "a proof from the device-generated Global table differs"
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = [pool.wait(t) for t in [pool.submit(pk, host2, publics) for _ in range(n_proofs)]]
            dt = time.perf_counter() - t0
            pool.close()
            gb = 4 * g.height * g.width
            out["staged_global_on_device"] = {
                "slots": 3, "ms_per_proof": 1e3 * dt / n_proofs, "tracegen_global_ms": min(tms), "global_rows": g.height, "global_events": n_ev,
                "device_table_equals_host_trace": same, "host_trace_bytes_per_proof": host_bytes - gb,
                "ms_per_proof_plus_tracegen": 1e3 * dt / n_proofs + min(tms),
                "note": "every trace but Global staged from pinned host memory, Global generated on the device from its %d events "
                        "(the tracegen time is listed beside the pipeline's; run back to back it is the upper bound of their sum)" % n_ev}
            del host2, table
        except Exception as e:                        # an extra: never the reason a line is missing
            print("bench.py: staged_global_on_device not measured: %r" % (e,), file=sys.stderr)
    del host
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fibonacci", choices=["real", "core", "precompile"] + list(PROGRAMS),
                    help="fibonacci | loop | keccak: a full core shard of that guest of the reference's perf harness (bench/program_shard.py; "
                         "value = cycles/s); real: the RISC-V chips at the recorded core shard's heights (bench/core_real.py); core: the synthetic "
                         "core-shaped shard of rounds 1-3 (bench/core_shard.py); precompile: a Keccak precompile shard, 82 %% of its "
                         "area in the 2,640-column KeccakPermute chip (bench/precompile_shard.py)")
    ap.add_argument("--scale-log2", type=int, default=0, help="prove a shard of area CORE >> 2k (testing aid; the bench line is k = 0)")
    ap.add_argument("--cpu-sample-scale-log2", type=int, default=1, help="the CPU baseline proves a shard of CORE >> 2k cells (default 1/4 of CORE; the 1/16 sample rides along)")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--verify-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-verify", action="store_true", help="skip the (untimed) verification of the last timed proof")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even at --gpus 1 (the N = 1 line then runs the same barrier / "
                         "max-over-ranks collectives as N = 8; RCCL when --backend nccl)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-program-run", action="store_true", help="skip the whole-run extra (every shard of a multi-shard execution, ~40 s)")
    ap.add_argument("--program-run-cycles", type=int, default=26_000_000, help="cycles of the whole-run extra (3 core shards of fibonacci + the memory shard)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (2-in-flight, commit-only, CPU baseline)")
    ap.add_argument("--mode", default="shard", choices=["shard", "tree"],
                    help="shard (default): every rank proves its own shard, weak scaling. tree: BASELINE config 5's shape — "
                         "--tree-leaves core shards striped over the ranks, their proofs reduced to ONE root through the recursion "
                         "compress tree (shards.reduce_tree + recursion_combine: every node a ShardProof of the recursion machine), "
                         "leaves + tree inside the timed region")
    ap.add_argument("--tree-leaves", type=int, default=0, help="leaves of --mode tree (default 2 x ranks + 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL; gloo lets "
                         "several ranks share one GPU when testing the N > 1 path on a 1-GPU box)")
    args = ap.parse_args()
    if args.cpu_baseline_child is not None:
        return cpu_baseline_child(args.cpu_baseline_child)
    if args.verify_child is not None:
        return verify_child(args.verify_child)

    # ONE JSON line on stdout, whatever the libraries print: RCCL writes "Librccl path : ..." to fd 1 when a communicator is
    # torn down. Keep the real stdout aside and point fd 1 at stderr for the rest of the run.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:                                       # plain `python bench.py --force-dist`: a one-rank rendezvous
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    from sp1_amd import api, shards
    lib = api._L()
    k = args.scale_log2
    kind = args.workload
    L, lsh = max(22 - k, 17 if kind in ("real", "precompile") + PROGRAMS else 0), 21 - k   # the Range table of the real machines has 2^17 rows
    chips, meta = build_workload(kind, k, L, 42 + rank)                  # every rank proves its own shard
    names = [c[0].name for c in chips]
    area = meta["area_cells"]
    jp = api.JaggedProver(L, lsh, 32, 2)
    prep_commit, prep_data = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])   # setup (the proving key)
    torch.cuda.synchronize()

    last_state = [None]
    publics = meta["publics"] if "publics" in meta else publics_of(kind)   # a real program's shard carries its own public values
    if args.mode == "tree":
        out = tree_mode(api, shards, dist, torch, args, rank, world, use_dist, chips, meta, prep_commit, prep_data, L, lsh, publics, kind, names)
        if rank == 0:
            result_out.write(json.dumps(out) + "\n")
            result_out.flush()
        if use_dist:
            dist.destroy_process_group()
        return

    def step(stream=None):
        ch = api.DuplexChallenger()
        ch.observe(prep_commit)                                              # stands for vk.observe_into
        blob = api.prove_shard(chips, publics, prep_data, L, lsh, 32, ch, stream=stream)
        last_state[0] = ch
        return blob

    for _ in range(args.warmup):
        step()
    if args.warmup:
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    if os.environ.get("SP1HIP_BENCH_PMC_MARK") == "1":       # bench/profile_bench_pmc.py: counters are kept from this kernel on
        n_cal = 1 << 28                                      # (a calibration kernel with a known byte count: 1 GiB each way)
        cal = torch.zeros(n_cal, dtype=torch.int32, device="cuda")
        api.check(lib.sp1hip_to_monty(api._dptr(cal), n_cal, api._stream_ptr()))
        torch.cuda.synchronize()
        del cal
    api.check(lib.sp1hip_timers_reset())
    api.check(lib.sp1hip_timers_enable(1))
    cpu0 = time.process_time()                               # user + system time of every thread of this process
    thr0 = thread_cpu_times()
    t0 = time.perf_counter()
    proof = None
    for _ in range(args.steps):
        proof = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host_cpu_ms = 1e3 * (time.process_time() - cpu0) / args.steps
    thr1 = thread_cpu_times()
    # where the host CPU time goes: the calling thread (transcript + waiting for hand-overs) against everything else (the HIP
    # runtime's own threads, helpers), ms per proof
    me = threading.get_native_id()
    host_cpu_by_thread = {"caller": round(1e3 * (thr1.get(me, (0, ""))[0] - thr0.get(me, (0, ""))[0]) / args.steps, 2)}
    others = sorted(((1e3 * (t - thr0.get(tid, (0, ""))[0]) / args.steps, comm, tid) for tid, (t, comm) in thr1.items() if tid != me), reverse=True)
    for rank_, (d, comm, tid) in enumerate(others):
        if d >= 0.05:                                        # the busiest other threads one by one (the HIP runtime's are all named like the process)
            key = "%s#%d" % (comm, rank_) if rank_ < 4 else comm + "#rest"
            host_cpu_by_thread[key] = round(host_cpu_by_thread.get(key, 0) + d, 2)
    host_cpu_by_thread["threads"] = len(thr1)
    api.check(lib.sp1hip_timers_enable(0))
    tl = {name: timers_read(api, name) for name in TIMERS}
    dt = shards.max_over_ranks(dt)                 # shards are striped one per rank: no data-path collective
    timed_state = last_state[0].state()

    # the pinned verifier on the LAST TIMED proof (rank 0, untimed): a line without `verified: true` is not printed
    verified = None
    if rank == 0 and not args.no_verify:
        verified = verify_proof(kind, proof, prep_commit, timed_state, L, lsh, names)
        if not verified:
            raise SystemExit("bench.py: the pinned verifier REJECTED the timed proof; no result line")

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:      # untimed extras; N > 1 runs measure scaling only
        extras["in_flight"] = in_flight(api, chips, area, L, lsh, max(4, args.steps), publics, meta.get("alu_events"))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            _, sd = jp.commit_multilinears([c[2] for c in chips])
            del sd
        torch.cuda.synchronize()
        extras["commit_only"] = {"ms": 1e3 * (time.perf_counter() - t1) / 3, "cells": sum(c[2].height * c[2].width for c in chips)}
        # the RS encode of the shard's stacked columns ALONE (in the proof it runs under the leaf hash): batches of 32 columns of
        # height 2^lsh, blowup 4, the same three passes
        try:
            S_all = -(-sum(c[2].height * c[2].width for c in chips) // (1 << lsh)) + 1
            cols = api.ColMajor(api.device_words((1 << lsh) * 32), 1 << lsh, 32)
            cw = api.device_words((4 << lsh) * 32)
            n_batches = -(-S_all // 32)
            for rep in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n_batches):
                    api.check(lib.sp1hip_rs_encode_batch(api._dptr(cw), api._dptr(cols.words), lsh, 2, 32, api._stream_ptr()))
                torch.cuda.synchronize()
                extras["rs_encode_alone_ms"] = 1e3 * (time.perf_counter() - t1)
            del cols, cw
        except Exception as e:                                    # an extra, never the reason a line is missing
            extras["rs_encode_alone_ms"] = None
            print("bench.py: rs_encode alone not measured: %r" % (e,), file=sys.stderr)
        # host CPU per proof with the event timers OFF (the timed loop above brackets every kernel with events for the roofline
        # object, which is work for the HIP runtime's own threads): what a production caller pays
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            c0, th0, t1 = time.process_time(), thread_cpu_times(), time.perf_counter()
            n_u = max(8, args.steps // 2)
            for _ in range(n_u):
                step()
            torch.cuda.synchronize()
            me_u, th1 = threading.get_native_id(), thread_cpu_times()
            extras["host_cpu_untimed"] = {
                "proofs": n_u, "ms_per_proof": 1e3 * (time.perf_counter() - t1) / n_u,
                "host_cpu_ms_per_proof": 1e3 * (time.process_time() - c0) / n_u,
                "caller_ms": round(1e3 * (th1.get(me_u, (0, ""))[0] - th0.get(me_u, (0, ""))[0]) / n_u, 2),
                "other_threads_ms": round(sum(1e3 * (t - th0.get(tid, (0, ""))[0]) / n_u for tid, (t, _) in th1.items() if tid != me_u), 2),
                "threads": len(th1)}
        except Exception as e:
            print("bench.py: untimed host CPU not measured: %r" % (e,), file=sys.stderr)
        if kind in PROGRAMS and k == 0 and not args.no_program_run:
            # a WHOLE run under the driver's clock: every shard of a multi-shard fibonacci execution, one proving key
            try:
                import prove_program
                torch.cuda.empty_cache()
                pr, last, pr_commit, _ = prove_program.program_run(api, kind if kind != "rsp" else "fibonacci", args.program_run_cycles, L, lsh)
                pr["last_shard_verified"] = verify_proof(kind, last["proof"], pr_commit, None, L, lsh, last["names"], vk_words=last["vk_words"])
                extras["program_run"] = pr
            except Exception as e:
                print("bench.py: program_run not measured: %r" % (e,), file=sys.stderr)
        extras["real_machine"] = real_machine(api)
        if kind == "real":
            extras["synthetic_core_shaped"] = synthetic_core_shaped(api, k)
        if not args.no_cpu_baseline:
            extras["cpu_baseline"] = cpu_baseline(api, max(args.cpu_sample_scale_log2, k), kind)
    if use_dist:
        dist.barrier()

    if rank == 0:
        ms = {name: m / args.steps for name, (cnt, m) in tl.items() if cnt}
        launches = {name: cnt // args.steps for name, (cnt, m) in tl.items() if cnt}
        h, A_main = 1 << lsh, sum(c[2].height * c[2].width for c in chips)
        S_cols = -(-A_main // h) + 1
        N = 4 * h
        perms = N * (-(-S_cols // 8)) if S_cols else 0
        # kernel groups: (timers summed, algorithmic bytes per proof — SURVEY §8d)
        zc_bivariate = os.environ.get("SP1HIP_ZC_BIVARIATE", "1") != "0"
        groups = {
            "leaf_hash": (["leaf_hash"], 4 * N * S_cols + 32 * N),
            "rs_encode": (["ntt_pass0", "ntt_pass1", "ntt_pass2"], 4 * h * S_cols * 5),
            # bivariate (default): rounds 0 + 1 read the base traces once (4A), rounds >= 2 read extension tables of A/4, A/8 ...
            # rows (8A in total); the table updates: fix2 reads 4A and writes 4A, later updates read 8A and write 4A in total.
            # sequential (SP1HIP_ZC_BIVARIATE=0): round 0 reads 4A, rounds >= 1 16A; updates read 4A + 16A, write 8A + 8A
            "zerocheck_round": (["zerocheck_round"], (12 if zc_bivariate else 20) * area),
            "zerocheck_fix": (["zerocheck_fix"], (20 if zc_bivariate else 36) * area),
            "gkr_pass": (["gkr_pass_sum", "gkr_pass_fold_sum", "gkr_pass_fold"], 156 * meta["first_layer_entries"]),
            # round 6: the first layer and the two tree levels below it are ONE kernel — it writes level L (4 + 16 B per entry), L - 1
            # (32 B per pair) and L - 2 (32 B per quad) = 44 B per first-layer entry; the rest of the tree, two levels per launch, reads
            # 8 E (1 + 1/4 + ...) and writes 6 E (1 + 1/4 + ...) = 18.7 B per entry (SP1HIP_GKR_FUSED=0: 20 and 52)
            "gkr_first_layer": (["gkr_first_layer"], 44 * meta["first_layer_entries"]),
            "gkr_transition": (["gkr_transition"], 56 * meta["first_layer_entries"] // 3),
            "jagged_fold": (["jagged_round0_sum", "jagged_fold0_sum", "jagged_fold_sum"], 28 * area),
        }
        # PMC table of THIS round (bench/pmc_traffic.sh -> profiles/r05_traffic.json, r05_traffic_precompile.json): HBM bytes and
        # SQ_INSTS_VALU per proof for every kernel group; the roofline fractions below are (table or live value) / (live time) / peak
        # The numerators cannot go stale silently: the table records the SHAPE of the proof its counters were taken on
        # (trace cells, first-layer entries, launches per kernel group); a live run that differs is reported as `stale` with the reason.
        pmc, pmc_note, pmc_stale = {}, "no committed PMC table for this workload", None
        for fn in ("r06_traffic_%s.json" % kind, "r05_traffic_%s.json" % kind, "r05_traffic.json", "r05_traffic_precompile.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    tt = json.load(f)
                if k == 0 and tt.get("workload") == kind:
                    pmc, pmc_note = tt["kernels"], tt["source"] + " [profiles/%s]" % fn
                    why = []
                    shape = tt.get("shape")
                    if not shape:
                        why.append("the table carries no shape (made before round 6)")
                    else:
                        for key, live in (("area_cells", area), ("first_layer_entries", meta["first_layer_entries"])):
                            if shape.get(key) != live:
                                why.append("%s %s in the table, %s live" % (key, shape.get(key), live))
                        for g_, (tn_, _) in groups.items():
                            live_l = sum(launches.get(n_, 0) for n_ in tn_)
                            if live_l and shape.get("launches", {}).get(g_) != live_l:
                                why.append("%s: %s launches in the table, %d live" % (g_, shape.get("launches", {}).get(g_), live_l))
                    pmc_stale = why
                    break
            except (OSError, ValueError, KeyError):
                pass
        if os.environ.get("SP1HIP_BENCH_PMC_META"):          # bench/pmc_traffic.sh: the shape this run's counters belong to
            with open(os.environ["SP1HIP_BENCH_PMC_META"], "w") as f:
                json.dump({"workload": kind, "area_cells": area, "first_layer_entries": meta["first_layer_entries"],
                           "launches": {g_: sum(launches.get(n_, 0) for n_ in tn_) for g_, (tn_, _) in groups.items()}}, f)
        stages = {}
        for g, (tn, alg_bytes) in groups.items():
            g_ms = sum(ms.get(n, 0.0) for n in tn)
            if not g_ms:
                continue
            g_launch = sum(launches.get(n, 0) for n in tn)
            row = pmc.get(g, {})
            valu = row.get("valu_lane_insts_per_proof")
            hbm_frac = alg_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
            valu_frac = valu / (g_ms * 1e-3) / VALU_PEAK_MEASURED if valu else None
            stages[g] = {"ms": g_ms, "launches": g_launch, "algorithmic_bytes": alg_bytes, "hbm_frac": hbm_frac,
                         "hbm_bytes_pmc": row.get("hbm_bytes_per_proof"), "valu_lane_insts_pmc": valu,
                         "valu_frac_vs_measured_int_rate": valu_frac,
                         "valu_frac_vs_guide_issue_rate": valu / (g_ms * 1e-3) / VALU_PEAK_GUIDE if valu else None,
                         "bound": "valu" if (valu_frac or 0.0) > hbm_frac else "hbm"}
        if "leaf_hash" in stages:
            stages["leaf_hash"]["permutations"] = perms
            if stages["leaf_hash"]["valu_lane_insts_pmc"]:
                stages["leaf_hash"]["valu_insts_per_permutation"] = stages["leaf_hash"]["valu_lane_insts_pmc"] / perms
        stages["timed_kernels_ms"] = sum(v for n, v in ms.items() if not n.startswith("stage_"))
        # Windows: the commit's two big kernel groups run CONCURRENTLY (the encode of batch k + 1 on a side stream under the leaf
        # hash of batch k), so each group's own launch time above is stretched by the other; what the pair achieves is priced
        # against the stage's window (the library's `stage_commit` timer on the caller's stream), and the whole proof against
        # the step (every kernel of the library in the PMC passes: `all_kernels`)
        windows = {}
        if ms.get("stage_commit") and "leaf_hash" in stages and "rs_encode" in stages:
            w_ms = ms["stage_commit"]
            v = (stages["leaf_hash"]["valu_lane_insts_pmc"] or 0) + (stages["rs_encode"]["valu_lane_insts_pmc"] or 0)
            b = (stages["leaf_hash"]["hbm_bytes_pmc"] or 0) + (stages["rs_encode"]["hbm_bytes_pmc"] or 0)
            windows["commit"] = {"ms": w_ms, "kernels": "leaf_hash + rs_encode (the tree's compress layers not counted: a lower bound)",
                                 "valu_frac_vs_measured_int_rate": v / (w_ms * 1e-3) / VALU_PEAK_MEASURED if v else None,
                                 "hbm_frac_pmc_bytes": b / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if b else None}
        for st in ("logup_gkr", "zerocheck", "evaluation_proof"):
            if ms.get("stage_" + st):
                windows.setdefault(st, {})["ms"] = ms["stage_" + st]
        allk = pmc.get("all_kernels")
        if allk:
            step_ms = 1e3 * dt / args.steps
            windows["whole_proof"] = {"ms": step_ms, "launches": allk["launches_per_proof"],
                                      "valu_frac_vs_measured_int_rate": allk["valu_lane_insts_per_proof"] / (step_ms * 1e-3) / VALU_PEAK_MEASURED,
                                      "hbm_frac_pmc_bytes": allk["hbm_bytes_per_proof"] / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                      "valu_lane_insts_pmc": allk["valu_lane_insts_per_proof"], "hbm_bytes_pmc": allk["hbm_bytes_per_proof"]}
        stages["windows"] = windows
        # the dominant kernel group of the step, by live-measured launch time over ALL timed kernels
        dom = max((g for g in groups if g in stages), key=lambda g: stages[g]["ms"])
        overlapped = None
        if dom in ("rs_encode", "leaf_hash") and "commit" in windows and "leaf_hash" in stages and "rs_encode" in stages:
            # the two commit groups run CONCURRENTLY (encode of batch k + 1 under the leaf hash of batch k): their summed launch
            # times are stretched by each other. The window's largest consumer is the one with more of its binding resource
            # (VALU lane-instructions from the PMC table, else its time alone); the other is reported beside it
            big = max(("leaf_hash", "rs_encode"), key=lambda g: stages[g]["valu_lane_insts_pmc"] or stages[g]["ms"])
            other = "rs_encode" if big == "leaf_hash" else "leaf_hash"
            dom = big
            overlapped = {"group": other, "in_step_ms": stages[other]["ms"], "alone_ms": extras.get("rs_encode_alone_ms") if other == "rs_encode" else None,
                          "window": "commit", "window_ms": windows["commit"]["ms"],
                          "window_valu_frac": windows["commit"]["valu_frac_vs_measured_int_rate"]}
        d = stages[dom]
        dom_ms, dom_launches, alg_dom = d["ms"], d["launches"], groups[dom][1]
        ms_per_step = 1e3 * dt / args.steps
        if d["bound"] == "valu":
            achieved, peak, unit = d["valu_lane_insts_pmc"] / (dom_ms * 1e-3) / 1e12, VALU_PEAK_MEASURED / 1e12, "T lane-inst/s"
        else:
            achieved, peak, unit = alg_dom / (dom_ms * 1e-3) / 1e9, HBM_PEAK_GBPS, "GB/s"
        traffic = d["hbm_bytes_pmc"] / dom_launches if d["hbm_bytes_pmc"] else None
        # (the essentials first: the driver's parse truncates long strings)
        if kind == "real":
            workload_text = ("core shard, rv64im machine: the %d chips of the core shape cluster (%d at height zero), %d constraints, %d interactions, "
                             "heights of the reference's recorded core shard 0, %d instructions executed"
                             % (len(meta["real_chips"]), len(meta["empty_chips"]), meta["constraints"], meta["interactions"],
                                meta["instructions_executed"]))
        elif kind == "precompile":
            workload_text = ("precompile shard: the %d chips of the Keccak shape cluster (%s), %d constraints, %d interactions"
                             % (len(meta["real_chips"]), ", ".join(meta["real_chips"]), meta["constraints"], meta["interactions"]))
        elif kind in PROGRAMS:
            workload_text = ("core shard %d of the reference's `%s` guest ELF (%d cycles = executed rv64im instructions, %.1f cells per cycle): "
                             "the %d chips of the reference's core shape cluster (%d without events, at height zero), %d constraints, %d interactions, "
                             "the shard's own public values (eval_public_values closes its buses)"
                             % (meta["shard_index"], kind, meta["cycles"], meta["cells_per_cycle"], len(meta["real_chips"]),
                                len(meta["empty_chips"]), meta["constraints"], meta["interactions"]))
        else:
            workload_text = ("core-shaped SYNTHETIC shard: %d chips, %d constraints, %d interactions"
                             % (meta["chips"], meta["constraints"], meta["interactions"]))
        cells_per_s = world * args.steps * area / dt
        cycles = meta.get("cycles", meta.get("instructions_executed"))
        if kind in PROGRAMS:            # BASELINE.json's metric, on the guest it names: RISC-V cycles proved per second
            headline = {"metric": "RISC-V cycles proved/sec (core shard prove): whole ShardProof of a full core shard of the reference's "
                                  "`%s` guest, see config" % kind,
                        "value": world * args.steps * meta["cycles"] / dt, "unit": "cycles/s"}
        else:
            headline = {"metric": "core shard prove throughput: trace cells proved/sec (whole ShardProof of the RISC-V core machine, see config)",
                        "value": cells_per_s, "unit": "cells/s"}
        out = {
            **headline, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (KoalaBear Montgomery words, exact integer arithmetic)", "data": ("synthetic: traces of an rv64im test program executed by this repository's executor"
                     if kind == "real" else "synthetic: seeded traces of real chips" if kind == "precompile"
                     else "the reference's guest binary (bench/programs/%s.elf.gz) on a synthetic input of sp1-gpu perf's form, executed by this "
                          "repository's rv64im executor; the shard is the reference's machine (RiscvAir core cluster + eval_public_values)" % kind if kind in PROGRAMS else "synthetic"),
            "proofs_per_s": world * args.steps / dt, "cells_per_s": cells_per_s,
            "riscv_instructions_per_s": world * args.steps * cycles / dt if cycles else None,
            "config": {"workload": workload_text + "; %d cells%s, L %d, stack 2^%d, blowup 4, 124 queries, 16-bit PoW; one whole ShardProof "
                                   "(sp1hip_prove_shard), traces resident in HBM" % (area, "" if k == 0 else " (scale 4^-%d)" % k, L, lsh),
                       "area_cells": area, "first_layer_entries": meta["first_layer_entries"], "proof_bytes": len(proof),
                       "instructions_executed": cycles, "program": meta.get("program"), "shard_index": meta.get("shard_index"),
                       "parallelism": "independent shards, one per GPU" + (" (rank r proves shard r of one execution)" if kind in PROGRAMS else "")},
            "roofline": {"bound": d["bound"], "kernel": dom, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_over_algorithmic": (traffic / (alg_dom / dom_launches)) if traffic else None,
                         "pmc_source": pmc_note, "stale": bool(pmc_stale) if pmc_stale is not None else None, "stale_because": pmc_stale or None,
                         "hbm_frac": d["hbm_frac"], "valu_frac": d["valu_frac_vs_measured_int_rate"],
                         "avg_launch_ms": dom_ms / dom_launches, "launches_per_step": dom_launches, "overlapped_with": overlapped,
                         "algorithmic_bytes_per_launch": alg_dom // dom_launches,
                         "note": "dominant kernel group of the step by live launch time (HIP events on the launch stream, all timed "
                                 "kernels considered); bound = whichever of algorithmic-bytes / time / 8 TB/s and SQ_INSTS_VALU x 64 / "
                                 "time / 39.3e12 is larger; VALU instructions and HBM bytes from the committed PMC passes of this "
                                 "round; peaks: HBM 8 TB/s (guide), VALU 256 CU x 64 lanes x 2.4 GHz (measured integer rate; the "
                                 "guide's packed-issue figure is twice that and is reported beside it in `stages`)",
                         "stages": stages},
            "cpu_baseline": extras.get("cpu_baseline"),
            "verified": verified,
            "core_real_chips": ({"real_chips": meta["real_chips"], "synthetic_chips": meta["synthetic_chips"],
                                 "real_area_cells": meta["real_area_cells"], "ms_per_proof": ms_per_step,
                                 "zerocheck_round_ms": ms.get("zerocheck_round"), "zerocheck_stage_ms": ms.get("stage_zerocheck"),
                                 "wide_chip_area_fraction": meta.get("wide_chip_area_fraction"), "per_chip": meta["per_chip"]}
                                if kind in ("real", "precompile") + PROGRAMS else None),
            "host_threads": lib.sp1hip_host_threads(), "host_cpu_ms_per_proof": host_cpu_ms, "host_cpu_ms_by_thread": host_cpu_by_thread,
            "host_wait": os.environ.get("SP1HIP_WAIT", "predict"), "host_cpu_untimed": extras.get("host_cpu_untimed"),
            "dist": {"initialised": use_dist, "backend": args.backend if use_dist else None},
            "program_run": extras.get("program_run"),
            "real_machine": extras.get("real_machine"),
            "synthetic_core_shaped": extras.get("synthetic_core_shaped"),
            "in_flight": extras.get("in_flight"),
            # the best of the in-flight runs, in the unit of `value`
            "value_pipelined": (max((v["cells_per_s"] for v in extras["in_flight"]["slots"].values()), default=None)
                                * (meta["cycles"] / area if kind in PROGRAMS else 1.0)) if extras.get("in_flight") else None,
            "commit_only": extras.get("commit_only"),
        }
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
