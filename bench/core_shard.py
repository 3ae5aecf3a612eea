"""A synthetic core shard SHAPED LIKE A REAL ONE: 33 chips whose widths and constraint counts are those
of RISC-V chips (rv64im_costs.json / rv64im_complexity.json), with the 730 interactions and the row counts of a recorded
core shard (layer_workloads.json) — bench/core_shape.json, derived by bench/make_core_shape.py. Row counts are scaled by
one common factor to the requested trace area (CORE = 2^28 + 2^27 cells).

The chips' real constraint polynomials need the Rust exporter; what is synthesised here, per chip, is a degree-3
constraint system with the real NUMBER of constraints over the real NUMBER of columns, satisfied by a random trace
generated on the device, and lookups that balance:
  columns come in quads (s, x, y, z) with s boolean and z = x y; constraint j of a quad is m_j * I_j with
  I_j in {x y - z, s (s - 1)} and m_j in {1, s, x, y, s', x', y', s - 1, ...} (primes: the next quad), so a chip has
  up to 16 distinct degree-2/3 constraints per quad — about 2 multiplications each after sharing sub-expressions, the
  density of the reference's ALU / memory chips;
  interaction 2k is a send and 2k + 1 a receive of the same 6-tuple of columns with multiplicity s (odd one out:
  multiplicity 0), so LogUp balances inside every chip, as 730 first-layer interactions over the real row counts.
"""
import json
import os

import torch

from sp1_amd import api
from sp1_amd.air import AirProgram, InteractionProgram, VCol

P = api.P
R_INV = pow(1 << 32, -1, P)
R1 = (1 << 32) % P


def load_shape():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "core_shape.json")) as f:
        return json.load(f)


def chip_programs(name, width, prep_width, n_constraints, n_interactions):
    air = AirProgram(name, width, prep_width, cse=True)
    quads = width // 4
    S, X, Y, Z = (lambda q: air.main(4 * q)), (lambda q: air.main(4 * q + 1)), (lambda q: air.main(4 * q + 2)), (lambda q: air.main(4 * q + 3))
    made = 0
    variant = 0
    while made < n_constraints and quads:
        for q in range(quads):
            if made == n_constraints:
                break
            n = (q + 1) % quads
            ident = (X(q) * Y(q) - Z(q)) if variant % 2 == 0 else S(q) * (S(q) - 1)
            mults = [None, S(q), X(q), Y(q), S(n), X(n), Y(n), S(q) - 1]
            m = mults[(variant // 2) % len(mults)]
            air.assert_zero(ident if m is None else m * ident)
            made += 1
        variant += 1
        assert variant < 16, "too many constraints for %d columns" % width
    inter = InteractionProgram(name, width, prep_width)
    cols = [VCol.main(c) for c in range(width)] + [VCol.prep(c) for c in range(prep_width)]
    for k in range(n_interactions // 2):
        vals = [cols[(7 * k + 3 * j + 1) % len(cols)] for j in range(6)]
        mult = VCol.main(4 * (k % quads)) if quads else VCol.const(0)
        kind = (1, 2, 5, 7, 8)[k % 5]
        inter.send(kind, vals, mult)
        inter.receive(kind, vals, mult)
    if n_interactions % 2:
        inter.send(5, [cols[j % len(cols)] for j in range(6)], VCol.const(0))
    assert inter.num_interactions == n_interactions
    return air, inter


def chip_trace(rows, width, gen):
    """[rows x width] column-major Montgomery words: quads (s, x, y, z = x y), leftover columns random."""
    out = torch.empty(rows * width, dtype=torch.int32, device="cuda")
    v = out.view(width, rows)
    quads = width // 4
    if quads:
        v[0:4 * quads:4] = torch.randint(0, 2, (quads, rows), dtype=torch.int32, device="cuda", generator=gen) * R1
        x = torch.randint(0, P, (quads, rows), dtype=torch.int64, device="cuda", generator=gen)
        y = torch.randint(0, P, (quads, rows), dtype=torch.int64, device="cuda", generator=gen)
        v[1:4 * quads:4] = x.to(torch.int32)
        v[2:4 * quads:4] = y.to(torch.int32)
        v[3:4 * quads:4] = ((x * y % P) * R_INV % P).to(torch.int32)          # Montgomery product of the two words
    if width > 4 * quads:
        v[4 * quads:] = torch.randint(0, P, (width - 4 * quads, rows), dtype=torch.int32, device="cuda", generator=gen)
    return api.ColMajor(out, rows, width)


def build_core_shard(area_target=(1 << 28) + (1 << 27), max_log_row_count=22, seed=42):
    """Returns (chips [(AirProgram, InteractionProgram, main, prep or None)] in name order, meta)."""
    shape = load_shape()
    recorded = sum(c["rows"] * (c["width"] + c["prep_width"]) for c in shape["chips"])
    scale = area_target / recorded
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    chips, area, n_int, n_con, entries = [], 0, 0, 0, 0
    for c in sorted(shape["chips"], key=lambda c: c["name"]):
        rows = min(max(8, int(round(c["rows"] * scale / 8)) * 8), 1 << max_log_row_count)
        air, inter = chip_programs(c["name"], c["width"], c["prep_width"], c["constraints"], c["interactions"])
        main = chip_trace(rows, c["width"], gen)
        prep = chip_trace(rows, c["prep_width"], gen) if c["prep_width"] else None
        chips.append((air, inter, main, prep))
        area += rows * (c["width"] + c["prep_width"])
        n_int += c["interactions"]
        n_con += c["constraints"]
        entries += rows * c["interactions"]
    meta = {"chips": len(chips), "interactions": n_int, "constraints": n_con, "area_cells": area, "first_layer_entries": entries,
            "row_scale": round(scale, 4), "shape_source": shape["source"]}
    return chips, meta
