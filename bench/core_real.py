"""The core shard of REAL RISC-V chips: the rv64im chips transcribed in sp1_amd/machines/riscv.py at the
heights of the reference's recorded core shard 0 (sp1-gpu/crates/logup_gkr/layer_workloads.json, decoded by chip name in
riscv.RECORDED_ROWS), proved on traces that sp1_amd/machines/riscv_trace.py EXECUTES (a loop body run K = 13 times: the
recorded shard executes ~6.2e6 instructions of a 4.7e5-instruction program).

What is real: EVERY chip of that shard (round 5 added DivRem, SyscallInstrs and SyscallCore: riscv_more.py) —
the 25 instruction chips, MemoryLocal, MemoryBump, StateBump, Program, Byte, Range, SyscallCore and the septic-curve Global chip
(a Poseidon2 permutation + curve arithmetic per row: a third of the shard's cells) — constraints, interactions, and traces
with RISC-V semantics whose lookups balance against the shard's public values (`eval_public_values`,
sp1_amd/machines/public_values.py: since round 6 the record's own messages close the State / GlobalAccumulation buses; the two
synthetic closing chips of rounds 4-5 are gone and the shard is the reference's core shape cluster, chips without events at
height zero). What is synthetic is the PROGRAM: a random rv64im loop, not a guest. `real_global=False` swaps the Global chip for
a sink + a filler of its recorded shape (the round-4 intermediate workload; kept for A/B, not verifiable).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sp1_amd import api                                            # noqa: E402
from sp1_amd.machines import riscv as R, riscv_trace as RT         # noqa: E402

# 8 executions of the loop body: the body then holds enough distinct load/store positions that MemoryLocal and Global come out at
# 0.96x their recorded heights (13 gave 0.65x: fewer positions than recorded touched words); Program is 1.55x as tall in exchange.
K_ITER = 8
NOT_INSTRUCTIONS = ("Byte", "Range", "Program", "MemoryLocal", "MemoryBump", "StateBump", "Global", "SyscallCore", "SyscallInstrs")
PUBLIC_VALUES = 160          # SP1_PROOF_NUM_PV_ELTS (hypercube/src/air/public_values.rs:L19): SyscallInstrs reads commit / exit-code words
P = api.P


def recorded_counts(scale, K=K_ITER):
    """Loop-body positions per instruction chip so that K executions give `scale` x the recorded heights."""
    counts = {n: max(1, int(round(rows * scale / K))) for n, rows in R.RECORDED_ROWS.items() if n not in NOT_INSTRUCTIONS}
    # SyscallInstrs rows are ECALLs of the body (a quarter of them, in the recorded shard, have a table: SyscallCore)
    counts["Ecall"] = max(1, int(round(R.RECORDED_ROWS["SyscallInstrs"] * scale / K)))
    return counts


def to_col_major(t):
    """canonical int64 [rows, width] on the device -> api.ColMajor (Montgomery words)."""
    rows, width = t.shape
    m = ((t << 32) % P).to(torch.int32)
    return api.ColMajor(m.t().contiguous().view(-1), rows, width)


SYNTHETIC = ("GlobalSink", "GlobalFiller")


def machine_only(scale=1.0, seed=1, K=K_ITER, device="cuda", real_global=True):
    """(machine, tables, public values) of the executed loop at `scale` x the recorded heights."""
    counts = recorded_counts(scale, K)
    pages = max(4, int(600 * scale))
    return RT.generate(counts, K=K, seed=seed, mem_pages=(pages, pages), device=device, real_global=real_global)


def build_real_shard(scale=1.0, seed=1, K=K_ITER, real_global=True):
    """[(AirProgram, InteractionProgram, main ColMajor, prep ColMajor | None)] in chip-name order + meta, on cuda."""
    from core_shard import chip_programs, chip_trace
    machine, tabs, publics = machine_only(scale, seed, K, "cuda", real_global)
    chips = {a.name: (a, i, to_col_major(tabs[a.name][1]), to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
             for a, i in machine}
    real_area = sum(c[2].height * (c[2].width + (c[3].width if c[3] is not None else 0)) for n, c in chips.items()
                    if n not in SYNTHETIC)
    del tabs
    synthetic = [n for n in chips if n in SYNTHETIC]
    if not real_global:
        w, ncons, nint = R.RECORDED["Global"]
        rows = max(32, int(round(R.RECORDED_ROWS["Global"] * scale / 32)) * 32)
        air, inter = chip_programs("GlobalFiller", w, 0, ncons, nint)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        chips["GlobalFiller"] = (air, inter, chip_trace(rows, w, gen), None)
        synthetic.append("GlobalFiller")
    out = [chips[n] for n in sorted(chips)]
    area = sum(c[2].height * (c[2].width + (c[3].width if c[3] is not None else 0)) for c in out)
    per_chip = {a.name: {"rows": m.height, "columns": a.main_width + a.prep_width, "constraints": a.num_constraints,
                         "interactions": i.num_interactions, "instructions": len(a.instrs),
                         "instr_per_constraint": round(len(a.instrs) / a.num_constraints, 2) if a.num_constraints else None}
                for a, i, m, _ in out}
    meta = {"chips": len(out), "real_chips": sorted(n for n in chips if n not in synthetic), "synthetic_chips": synthetic,
            "empty_chips": sorted(c[0].name for c in out if c[2].height == 0), "publics": RT.to_monty_np(publics),
            "area_cells": area, "real_area_cells": real_area, "interactions": sum(c[1].num_interactions for c in out),
            "constraints": sum(c[0].num_constraints for c in out),
            "first_layer_entries": sum(c[2].height * c[1].num_interactions for c in out),
            "instructions_executed": K * sum(recorded_counts(scale, K).values()), "loop_iterations": K, "per_chip": per_chip}
    return out, meta


def programs_for(names):
    """(AirProgram, InteractionProgram) by chip name, for a verifier child that has no traces."""
    from core_shard import chip_programs
    out = []
    for n in names:
        if n == "GlobalSink":
            out.append(RT.global_sink_chip())
        elif n == "GlobalFiller":
            w, ncons, nint = R.RECORDED["Global"]
            out.append(chip_programs("GlobalFiller", w, 0, ncons, nint))
        else:
            out.append(R.chip(n))
    return out
