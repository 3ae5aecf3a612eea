"""A KECCAK_PERMUTE precompile shard: the chips a tendermint / rsp proof spends most of its precompile area in.
KeccakPermute (2,640 columns, 2,859 constraints, 24 rows per permutation), its controller, SyscallPrecompile, MemoryLocal, the
septic-curve Global chip, Program, Byte and Range — the reference's Keccak shape cluster, all REAL chips (sp1_amd/machines/
riscv_more.py, traces computed by riscv_more_trace.py), closed by the shard's public values. ~77k trace cells per permutation,
82 % of them in the wide chip; 5,120 permutations fill a shard of the core shard's area limit (2^28 + 2^27 cells)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sp1_amd.machines import riscv as R, riscv_more_trace as MT, riscv_trace as RT         # noqa: E402

FULL_EVENTS = 5120


def build_precompile_shard(n_events=FULL_EVENTS, seed=1, device="cuda"):
    """[(AirProgram, InteractionProgram, main ColMajor, prep ColMajor | None)] in chip-name order + meta."""
    from core_real import SYNTHETIC, to_col_major
    machine, tabs, publics = MT.precompile_shard(n_events, seed=seed, device=device)
    out = []
    for a, i in machine:
        prep, main = tabs[a.name]
        out.append((a, i, to_col_major(main), to_col_major(prep) if prep is not None else None))
        tabs[a.name] = None
    area = sum(c[2].height * (c[2].width + (c[3].width if c[3] is not None else 0)) for c in out)
    synthetic = [c[0].name for c in out if c[0].name in SYNTHETIC]
    wide = next(c for c in out if c[0].name == "KeccakPermute")
    per_chip = {a.name: {"rows": m.height, "columns": a.main_width + a.prep_width, "constraints": a.num_constraints,
                         "interactions": i.num_interactions, "instructions": len(a.instrs)} for a, i, m, _ in out}
    meta = {"chips": len(out), "real_chips": sorted(c[0].name for c in out if c[0].name not in synthetic), "synthetic_chips": synthetic,
            "area_cells": area, "real_area_cells": area - sum(c[2].height * c[2].width for c in out if c[0].name in synthetic),
            "interactions": sum(c[1].num_interactions for c in out), "constraints": sum(c[0].num_constraints for c in out),
            "first_layer_entries": sum(c[2].height * c[1].num_interactions for c in out), "keccak_permutations": n_events,
            "wide_chip_area_fraction": wide[2].height * wide[2].width / area, "per_chip": per_chip,
            "empty_chips": sorted(c[0].name for c in out if c[2].height == 0), "publics": RT.to_monty_np(publics)}
    return out, meta
