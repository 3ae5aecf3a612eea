"""GPU (-m gpu): device trace generation for the RISC-V instruction chips (sp1hip_tracegen_riscv_alu: Add, Addi, Sub, Addw, Subw,
Mul, ShiftRight, Branch) against the host traces of the same events (sp1_amd/machines/riscv_exec.py — whose fillers a second
independent reading agrees with cell for cell, tests/test_riscv_second_reading.py): every word of every column, padding rows
included; events of real guests and of hand-assembled programs that reach the corner cases (MULH / MULHSU signs, SRA / SRAW of
negative values, shifts by 0, signed and unsigned branches both ways, immediates, x0 operands); a shard proof made from the
device-generated tables equals the one made from the host tables."""
import os
import struct
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))
import rv_asm as A  # noqa: E402
from sp1_amd.machines import riscv as R  # noqa: E402
from sp1_amd.machines import riscv_exec as X  # noqa: E402
from sp1_amd.machines import riscv_trace as RT  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def test_table_widths_are_the_transcribed_chips(api):
    for name, kind in api.RISCV_ALU_CHIPS.items():
        assert api._L().sp1hip_tracegen_riscv_alu_width(kind) == R.chip(name)[0].main_width, name
    assert api._L().sp1hip_tracegen_riscv_alu_width(99) == -1
    assert set(api.RISCV_ALU_CHIPS) == set(X.ALU_TRACEGEN_CHIPS)


def _compare(api, ex, sh):
    import core_real
    machine, tabs, publics = X.shard_tables(ex, sh, device="cuda")
    ev = torch.as_tensor(sh.events, device="cuda")
    seen = {}
    for name in X.ALU_TRACEGEN_CHIPS:
        main = tabs[name][1]
        if main.shape[0] == 0:
            continue
        packed = X.pack_alu_events(ev, name)
        assert packed.shape[1] == api.ALU_EVENT_WORDS and 0 < packed.shape[0] <= main.shape[0]
        got = api.tracegen_riscv_alu(name, packed, main.shape[0])
        want = core_real.to_col_major(main)
        g, w = got.words.view(got.width, got.height), want.words.view(want.width, want.height)
        bad = (g != w).nonzero()
        lay = {v: k for k, v in R.chip(name)[0].layout.items()}
        assert bad.numel() == 0, (name, "first mismatch (column, row):", bad[0].tolist(), "column group", max((c for c in lay if c <= int(bad[0][0])), default=None) and lay[max(c for c in lay if c <= int(bad[0][0]))])
        seen[name] = (int(packed.shape[0]), got)
    return machine, tabs, publics, seen


def test_device_tables_of_the_fibonacci_guest_equal_the_host_traces(api):
    ex = X.Executor(X.guest_file("fibonacci.elf"), stdin=[struct.pack("<Q", 30000)])
    sh = ex.run_shard(1 << 20)
    assert sh.halted
    _, _, _, seen = _compare(api, ex, sh)
    assert {"Add", "Addi", "Sub", "Addw", "Mul", "ShiftRight", "Branch"} <= set(seen)
    assert sum(n for n, _ in seen.values()) > 0.98 * sh.events.shape[0]               # the loop's instructions are these seven chips'


def test_device_tables_on_corner_cases(api):
    """MUL / MULH / MULHU / MULHSU / MULW, SRL / SRA / SRLW / SRAW by 0, 1, 15, 16, 31, 32, 47, 63 (register and immediate), ADDW / SUBW
    overflows, every branch kind taken and not taken on signed / unsigned boundaries, rs1 = x0."""
    M64 = (1 << 64) - 1
    vals = [0, 1, M64, 1 << 63, (1 << 63) - 1, 0x8000_0000, 0x7FFF_FFFF, 0xFFFF_FFFF, 0x1234_5678_9ABC_DEF0, 0xFFFF_0000_FFFF_0001]
    prog = []
    for i, v in enumerate(vals):
        prog += A.li(5, v)
        for j, w in enumerate(vals[i % 3::3]):
            prog += A.li(6, w)
            for op in ("mul", "mulh", "mulhu", "mulhsu", "mulw", "add", "sub", "addw", "subw", "srl", "sra", "srlw", "sraw"):
                prog.append(A.enc(op, 7, 5, 6))
            for op in ("beq", "bne", "blt", "bge", "bltu", "bgeu"):
                prog.append(A.enc(op, 5, 6, 8))                             # taken or not, the next instruction is 8 bytes on ... (a nop between)
                prog.append(A.enc("addi", 0, 0, 0))
        for sh_ in (0, 1, 15, 16, 31, 32, 47, 63):
            prog += [A.enc("srli", 7, 5, sh_), A.enc("srai", 7, 5, sh_)]
            if sh_ < 32:
                prog += [A.enc("srliw", 7, 5, sh_), A.enc("sraiw", 7, 5, sh_)]
        prog += [A.enc("addi", 7, 5, -2048), A.enc("addi", 7, 5, 2047), A.enc("addiw", 7, 5, -1), A.enc("beq", 0, 0, 8), A.enc("addi", 0, 0, 0), A.enc("blt", 0, 5, 8),
                 A.enc("addi", 0, 0, 0)]
    ex = X.Executor(A.elf(prog + A.halt(0)), stdin=[])
    sh = ex.run_shard(1 << 20)
    assert sh.halted and sh.exit_code == 0
    _, _, _, seen = _compare(api, ex, sh)
    assert set(seen) >= {"Add", "Addi", "Sub", "Addw", "Subw", "Mul", "ShiftRight", "Branch"}


def test_shard_proof_from_device_generated_tables(api):
    import core_real
    ex = X.Executor(X.guest_file("fibonacci.elf"), stdin=[struct.pack("<Q", 200)])
    sh = ex.run_shard(1 << 20)
    machine, tabs, publics, seen = _compare(api, ex, sh)
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    L, lsh, batch = 17, 12, 8
    commit, prep = api.JaggedProver(L, lsh, batch, 1).commit_multilinears([d[3] for d in dev if d[3] is not None])

    def prove(chips):
        ch = api.DuplexChallenger()
        ch.observe(commit)
        return api.prove_shard(chips, RT.to_monty_np(publics), prep, L, lsh, batch, ch, 1, 5, 4)
    want = prove(dev)
    dev2 = [(a, i, seen[a.name][1] if a.name in seen else m, p) for a, i, m, p in dev]
    assert prove(dev2) == want
