"""Core shards of REAL guest programs (VERDICT r4 "real-program workloads: no"): the reference's own benchmark guests
(bench/programs/*.elf.gz, see the README there) executed by the rv64im executor of libsp1hip.so and traced by
sp1_amd/machines/riscv_exec.py — every chip a real chip of the RISC-V machine, the events those of the guest's execution.

The default is what BASELINE.json's metric is quoted on: a full core shard of `fibonacci` ("RISC-V cycles proved / second").
Shard size: the reference cuts a shard when its estimated trace area reaches ELEMENT_THRESHOLD = 2^28 + 2^27 cells or a
chip reaches 2^22 rows (/root/reference/crates/core/executor/src/opts.rs:L12-L14); fibonacci's loop costs 45.6 cells per
cycle (9 instructions per iteration: Add, Addi, Sub, Mul x 2, Addw, ShiftRight, Branch, ...), so the area binds first, at
8.8e6 cycles. `FULL_CYCLES = 2^23` (8.39e6 cycles, 3.9e8 cells incl. the fixed tables) is the power of two below it.
Rank r of a multi-GPU run proves shard r of the same execution: independent shards, one per GPU, no data-path collective."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL_CYCLES = 1 << 23
# the other guests: the power of two below the cycle count at which THEIR mix reaches the area threshold (measured cells per cycle:
# loop 37, keccak 61, poseidon2 65, sha2 113)
FULL_CYCLES_OF = {"fibonacci": 1 << 23, "loop": 1 << 23, "keccak": 1 << 22, "poseidon2": 1 << 22, "sha2": 1 << 21, "rsp": 1 << 23}
CYCLES_PER_UNIT = {"fibonacci": 9, "loop": 4, "keccak": 7, "sha2": 4, "poseidon2": 9}   # measured: cycles per loop iteration / per input byte


def stdin_of(program, cycles):
    """The input of sp1-gpu/crates/perf/src/lib.rs:L23-L45 sized so that the run lasts at least `cycles` cycles."""
    if program == "rsp":             # `write_vec(client_input)`: block 21740136 of the reference's perf inputs (lib.rs:L47-L52). The guest runs
        # 4.8e7 cycles (deserialisation, witness database: 5 full core shards) before its first hook — a hint computed outside the
        # VM (fd 20), which this executor does not implement: the shards before it are complete
        from sp1_amd.machines.riscv_exec import guest_file
        return [guest_file("rsp_input_21740136.bin")]
    n = cycles // CYCLES_PER_UNIT[program] + 1
    return [bytes(n)] if program in ("keccak", "sha2") else [struct.pack("<Q", n)]  # `write_vec(vec![0u8; n])` / `write(&n)`, n: usize


def build_program_shard(program="fibonacci", k=0, shard_index=0, device="cuda"):
    """[(AirProgram, InteractionProgram, main ColMajor, prep ColMajor | None)] in chip-name order + meta for core shard
    `shard_index` of `program` run with shards of FULL_CYCLES_OF[program] >> 2k cycles."""
    from core_real import SYNTHETIC, to_col_major
    from sp1_amd.machines import riscv_exec as X, riscv_trace as RT
    max_cycles = FULL_CYCLES_OF[program] >> (2 * k)
    elf = X.guest_file(program + ".elf")
    ex = X.Executor(elf, stdin=stdin_of(program, (shard_index + 1) * max_cycles + max_cycles // 8))
    prev = None
    for i in range(shard_index + 1):                         # the shards before this rank's run without keeping their events;
        shard = ex.run_shard(max_cycles, record=i == shard_index, copy=False)
        if i < shard_index:                                  # their public values chain into this shard's prev_* words
            prev = X.execution_public_values(shard, prev)
    assert shard.cycles == max_cycles and not shard.halted, "the run is too short for a full shard %d" % shard_index
    machine, tabs, publics = X.shard_tables(ex, shard, device, prev=prev)
    # the compact event records of the chips whose tables the device can generate itself (api.tracegen_riscv_alu): what a host
    # hands over instead of their tables — 88 B per instruction where the row is 120-328 B
    names_of = X.chip_of_events(shard.events)
    alu_events = {n: X.pack_alu_events(shard.events[np.nonzero(names_of == n)[0]]) for n in X.ALU_TRACEGEN_CHIPS if (names_of == n).any()}
    out = []
    for a, i in machine:
        prep, main = tabs[a.name]
        out.append((a, i, to_col_major(main), to_col_major(prep) if prep is not None else None))
        tabs[a.name] = None
    area = sum(c[2].height * (c[2].width + (c[3].width if c[3] is not None else 0)) for c in out)
    synthetic = [c[0].name for c in out if c[0].name in SYNTHETIC]
    per_chip = {a.name: {"rows": m.height, "columns": a.main_width + a.prep_width, "constraints": a.num_constraints,
                         "interactions": i.num_interactions, "instructions": len(a.instrs)} for a, i, m, _ in out}
    meta = {"chips": len(out), "real_chips": sorted(c[0].name for c in out if c[0].name not in synthetic), "synthetic_chips": synthetic,
            "empty_chips": sorted(c[0].name for c in out if c[2].height == 0),
            "area_cells": area, "real_area_cells": area - sum(c[2].height * c[2].width for c in out if c[0].name in synthetic),
            "interactions": sum(c[1].num_interactions for c in out), "constraints": sum(c[0].num_constraints for c in out),
            "first_layer_entries": sum(c[2].height * c[1].num_interactions for c in out),
            "program": program, "shard_index": shard_index, "cycles": shard.cycles, "clk": [shard.clk_start, shard.clk_end],
            "pc_start": shard.pc_start, "next_pc": shard.next_pc, "cells_per_cycle": area / shard.cycles,
            "publics": RT.to_monty_np(publics), "per_chip": per_chip,       # Montgomery words, like the tables
            "alu_events": alu_events}
    return out, meta
