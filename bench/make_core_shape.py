#!/usr/bin/env python3
"""Derive bench/core_shape.json — the shape of a REAL core shard — from data the reference tree holds (run in the
build container; /root/reference does not exist on the GPU box, so the result is committed):

  * /root/reference/sp1-gpu/crates/logup_gkr/layer_workloads.json: 119 recorded core shards; per shard the row count of
    every interaction (730 interactions, num_row_variables 21). Runs of equal row counts are the chips of the shard:
    shard 0 has 33 chips with (rows, #interactions) = (14216, 21), (362856, 17), ...
  * /root/reference/crates/core/executor/src/artifacts/rv64im_costs.json: columns per row of each of the 122 RISC-V chips;
  * /root/reference/crates/core/executor/src/artifacts/rv64im_complexity.json: `chip.num_constraints` of each chip
    (crates/core/machine/src/riscv/mod.rs:L1843-L1863).

The recording does not name the chips, so chips are PAIRED BY RANK: the 33 chips of the non-precompile RISC-V cluster
with the most columns get the interaction counts in descending order (wide chips have more lookups). Widths, constraint
counts, interaction counts and row counts are therefore each real; only their pairing is a heuristic.
"""
import json
import os
from itertools import groupby

REF = "/root/reference"
CORE_CHIPS = ["Add", "Addi", "Addw", "AluX0", "Bitwise", "Branch", "DivRem", "Global", "InstructionDecode", "InstructionFetch", "Jal",
              "Jalr", "LoadByte", "LoadDouble", "LoadHalf", "LoadWord", "LoadX0", "Lt", "MemoryBump", "MemoryLocal", "Mul", "Program",
              "Byte", "Range", "ShiftLeft", "ShiftRight", "StateBump", "StoreByte", "StoreDouble", "StoreHalf", "StoreWord", "Sub",
              "Subw", "SyscallCore", "SyscallInstrs", "UType"]
PREPROCESSED = {"Program": 10, "Byte": 8, "Range": 2}      # chips that also carry preprocessed columns (widths illustrative)


def main():
    costs = json.load(open(os.path.join(REF, "crates/core/executor/src/artifacts/rv64im_costs.json")))
    cplx = json.load(open(os.path.join(REF, "crates/core/executor/src/artifacts/rv64im_complexity.json")))
    shard = json.load(open(os.path.join(REF, "sp1-gpu/crates/logup_gkr/layer_workloads.json")))[0]
    groups = [(rows, len(list(g))) for rows, g in groupby(shard["interaction_row_counts"])]
    assert sum(n for _, n in groups) == 730 and shard["num_row_variables"] == 21
    chips = sorted(CORE_CHIPS, key=lambda n: -costs[n])[:len(groups)]
    by_int = sorted(range(len(groups)), key=lambda k: -groups[k][1])
    out = []
    for rank, k in enumerate(by_int):
        name = chips[rank]
        rows, n_int = groups[k]
        prep = PREPROCESSED.get(name, 0)
        out.append({"name": name, "width": costs[name] - prep, "prep_width": prep, "constraints": cplx[name], "interactions": n_int,
                    "rows": rows})
    out.sort(key=lambda c: c["name"])
    doc = {"source": "layer_workloads.json shard 0 (rows, interactions) x rv64im_costs.json (width) x rv64im_complexity.json (constraints); "
                     "paired by rank, see bench/make_core_shape.py",
           "num_row_variables": shard["num_row_variables"], "chips": out}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "core_shape.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    area = sum(c["rows"] * (c["width"] + c["prep_width"]) for c in out)
    print("wrote %s: %d chips, %d interactions, %d constraints, area %.3e cells as recorded" %
          (path, len(out), sum(c["interactions"] for c in out), sum(c["constraints"] for c in out), area))


if __name__ == "__main__":
    main()
