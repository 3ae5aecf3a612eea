"""Synthetic core-shard workload shared by bench/bench_shard.py and the full-size GPU parity test: chips with
degree-3 constraints and balanced lookups over random satisfying traces resident in HBM."""
import numpy as np
import torch

from sp1_amd import api
from sp1_amd.air import AirProgram


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


def wide_interactions(name, width, send, prep_width=0, lookups=True):
    """One lookup per 16 columns: the tuple (a, b, c) of the group's first quad with multiplicity d (boolean)."""
    from sp1_amd.air import InteractionProgram, VCol
    p = InteractionProgram(name, width, prep_width)
    for g in range(0, width // 4 if lookups else 0, 4):
        vals = [VCol.main(4 * g), VCol.main(4 * g + 1), VCol.main(4 * g + 2)]
        (p.send if send else p.receive)(5, vals, VCol.main(4 * g + 3))
    return p




def build_shard(L, lsh, area_target, seed=42):
    """Returns (chips [(AirProgram, InteractionProgram, main ColMajor, prep ColMajor or None)] in name order, the
    preprocessed ColMajor, shapes, area). Every shape appears twice, as a sender and as a receiver of the same tuples
    (same device trace), so the lookup argument balances and the proof is a valid one."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    shapes, area = [], 0
    widths = [8, 16, 32, 64, 100, 200, 400]
    k = 0
    while area < area_target:
        w = widths[k % len(widths)] // 4 * 4
        rows = (1 << (L - 1)) >> (k % 4)
        while 2 * rows * w > area_target - area and rows > 32:
            rows >>= 1
        shapes.append((rows, w))
        area += 2 * rows * w
        k += 1
    traces = [wide_trace(r, w, gen) for r, w in shapes]
    chips = []
    for i, ((r, w), t) in enumerate(zip(shapes, traces)):
        chips.append((wide_air(w), wide_interactions("R%02d" % i, w, False), t, None))
        chips.append((wide_air(w), wide_interactions("S%02d" % i, w, True), t, None))
    prep_rows = 1 << max(L - 6, 5)
    prep_air = AirProgram("Prep", 4, prep_width=2)
    prep_air.assert_zero(prep_air.prep(0) * (prep_air.main(3) * (prep_air.main(3) - 1)))
    prep_air.assert_eq(prep_air.main(2), prep_air.main(0) * prep_air.main(1))
    prep_main = wide_trace(prep_rows, 4, gen)
    prep_prep = api.ColMajor(torch.randint(0, api.P, (2 * prep_rows,), dtype=torch.int32, device="cuda", generator=gen), prep_rows, 2)
    chips.append((prep_air, wide_interactions("Prep", 4, True, prep_width=2, lookups=False), prep_main, prep_prep))
    for c in chips:
        c[0].name = c[1].name
    chips.sort(key=lambda c: c[1].name)
    return chips, prep_prep, shapes, area
