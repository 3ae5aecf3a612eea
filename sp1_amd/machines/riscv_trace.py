"""Satisfying traces for the transcribed RISC-V chips of `riscv.py`: a small rv64im EXECUTOR plus the chips' trace generation.

The reference fills its tables from the `ExecutionRecord` of its Rust executor (crates/core/executor) through each chip's
`generate_trace_into` / `event_to_row` (cited per chip below). That executor and real guest programs are Rust and out of
scope; what this module executes instead is a *synthetic but real* rv64im program — a loop body of random
instructions with true RISC-V semantics — and it fills every column from the executed values exactly as the reference's
`populate` functions would. Nothing is "made to fit": tests/machine_check.py then requires every constraint of every chip
to vanish on every row and the Byte / Memory / Program / State / Global buses to balance as multisets.

Program shape (vectorisable by construction, all tensors torch.int64, runs on CPU for tests and on the GPU for the bench):

    registers   x0 | B = x1..x8 scalars | PR = x9,x10 pointers into a read-only region | PS = x11,x12 pointers into a
                store region | D1 = x13..x20 | x21 link register (AUIPC) | D2 = x22..x31
    body        L instructions at pc_base + 4 i, executed K times (a loop: the last instruction is `jal x0, -4 (L - 1)`):
                  class A  sources in B u {x0}, destination in D1 (ALU ops, AUIPC, JAL) or PR / PS (LUI with a page of the region)
                  class B  sources in B u D1 u PR u PS, destination in D2 (ALU ops, loads through PR, JALR through the link register)
                  class S  no register destination: stores through PS, branches (taken branches jump to pc + 4)
                  tail     `addi xb, xb, step_b` for every scalar (their values are init_b + k step_b in iteration k), then the jump
    memory      loads read the read-only region (random initial image), stores write the store region; every touched word and
                register gets a MemoryLocal row (initial record at timestamp 0, final record = last access)

So the value an instruction reads is the value written by the statically known last writer of that register, in this or
the previous iteration — values come out of three vectorised passes (tail, A, B) instead of a sequential interpreter.
Timestamps: instruction n runs at clk0 + 8 n (CLK_INC), its accesses at +1 memory, +2 op_c, +3 op_b, +4 op_a
(MemoryAccessPosition); crossing a 2^24 boundary of the clock produces StateBump / MemoryBump rows like the reference
(adapter/bump.rs, memory/bump.rs), as does a carry out of the low pc limb.

Byte / Range / Program multiplicities are not re-derived chip by chip: they are COUNTED from the messages the chips' own
interaction programs send on the generated rows (every message is checked to be a row of the table it addresses).

What closes a shard is the reference's own record-level statement, `eval_public_values` (public_values.py): the shard's public
values send the initial CPU state and receive the final one, hold the two ends of the Global chip's accumulation chain, and
range-check their own limbs on the Byte bus — so a shard's tables are the chips of its shape cluster (riscv/mod.rs:L560-L803,
chips without events at height zero) and nothing else. (`real_global=False` swaps the Global chip for a synthetic `GlobalSink`
that only receives MemoryLocal's messages: a trace-generator test aid, never proved.)
"""
import numpy as np
import torch

from ..air import AirProgram, InteractionProgram, P, VCol
from . import riscv as R

I64 = torch.int64
MASK16, MASK32 = 0xFFFF, 0xFFFFFFFF
MIN64 = -(1 << 63)
POS_OFF = {"M": 1, "C": 2, "B": 3, "A": 4}
B_REGS, PR_REGS, PS_REGS = list(range(1, 9)), [9, 10], [11, 12]
D1_REGS, D2_REGS, LINK_REG = list(range(13, 20)), list(range(22, 31)), 21      # x20 / x31 carry syscall codes (ECALL_CODES)
PAGE = 4096

# instruction kinds: name -> (chip, opcode names)
ALU_KINDS = {
    "Add": ["ADD"], "Sub": ["SUB"], "Addi": ["ADDI"], "Bitwise": ["XOR", "OR", "AND"], "Lt": ["SLT", "SLTU"],
    "Mul": ["MUL", "MULH", "MULHU", "MULHSU", "MULW"], "ShiftLeft": ["SLL", "SLLW"], "ShiftRight": ["SRL", "SRA", "SRLW", "SRAW"],
    "Addw": ["ADDW"], "Subw": ["SUBW"], "DivRem": ["DIV", "DIVU", "REM", "REMU", "DIVW", "DIVUW", "REMW", "REMUW"],
}
# ECALLs of the loop body (kind "Ecall"): `addi x31, x0 | x20, code` (+ `lui x20, hi` for a 3-byte code) then `ecall` with
# op_a = x31, op_b = x10, op_c = x11 — the reference's decoder always names x5 / x10 / x11 (core/executor: Instruction::new(ECALL,
# 5, 10, 11)), the SyscallInstrs AIR takes the registers from the program table like any R-type row. Codes that neither change
# op_a nor read public values: WRITE, EXIT_UNCONSTRAINED, HINT_LEN (no table), KECCAK_PERMUTE / SHA_EXTEND (own table: the row
# is also a SyscallCore row and a Global send; the precompile itself lives in another shard).
ECALL_CODES = [0x02, 0x04, 0xF0, 0x00_01_01_09, 0x00_30_01_05]
SYSCALL_REG, SYSCALL_HI_REG = 31, 20
ECALL_EXTRA_CLK = 256                                                          # syscall/instructions/air.rs:L66-L72
IMM_CAPABLE = {"Bitwise", "Lt", "ShiftLeft", "ShiftRight", "Addw"}           # ALUTypeReader chips: op_c may be an immediate
LOAD_KINDS = {"LoadByte": ["LB", "LBU"], "LoadHalf": ["LH", "LHU"], "LoadWord": ["LW", "LWU"], "LoadDouble": ["LD"],
              "LoadX0": ["LB", "LBU", "LH", "LHU", "LW", "LWU", "LD"]}          # LoadX0: any load whose destination is x0
STORE_KINDS = {"StoreByte": ["SB"], "StoreHalf": ["SH"], "StoreWord": ["SW"], "StoreDouble": ["SD"]}
BRANCH_OPS = ["BEQ", "BNE", "BLT", "BGE", "BLTU", "BGEU"]
ACCESS_BYTES = {"LB": 1, "LBU": 1, "LH": 2, "LHU": 2, "LW": 4, "LWU": 4, "LD": 8, "SB": 1, "SH": 2, "SW": 4, "SD": 8}
OPC = R.OPC


def pad32(n):
    return max(-(-n // 32) * 32, 32)


def fpow(x, e):
    r = torch.ones_like(x)
    while e:
        if e & 1:
            r = r * x % P
        x = x * x % P
        e >>= 1
    return r


def finv(x):
    return fpow(x % P, P - 2)


def limbs16(x):
    """[n] int64 (two's complement 64-bit values) -> [n, 4] 16-bit limbs."""
    return torch.stack([(x >> (16 * i)) & MASK16 for i in range(4)], dim=1)


def bytes8(x):
    return torch.stack([(x >> (8 * i)) & 0xFF for i in range(8)], dim=1)


def sext32(x):
    x = x & MASK32
    return x - ((x >> 31) << 32)


def ult(a, b):
    return (a ^ MIN64) < (b ^ MIN64)


def srl(a, s):
    """logical right shift of int64 by s in [0, 63]"""
    mask = torch.where(s == 0, torch.full_like(a, -1), (torch.ones_like(a) << (64 - s).clamp(max=63)) - 1)
    return (a >> s) & mask


class Body:
    """The static loop body. Arrays over body positions."""

    def __init__(self, counts, rng, pc_base, mem_pages):
        kinds = []
        for name, n in counts.items():
            if name in ALU_KINDS or name == "Jalr" or name in LOAD_KINDS or name in STORE_KINDS or name in ("Branch", "Jal", "UType", "Ecall"):
                kinds += [(name,)] * n
            else:
                raise KeyError(name)
        order = rng.permutation(len(kinds))
        op, chip, rd, rs1, rs2, imm, has_imm = [], [], [], [], [], [], []

        def emit(chip_, op_, rd_, rs1_, rs2_, imm_, has_imm_):
            chip.append(chip_); op.append(OPC[op_]); rd.append(rd_); rs1.append(rs1_); rs2.append(rs2_); imm.append(imm_)
            has_imm.append(has_imm_)

        pick = lambda lst: lst[int(rng.integers(len(lst)))]
        r_pages, s_pages = mem_pages
        self.r_base, self.s_base = 0x100000, 0x100000 + (r_pages + 1) * PAGE
        self.r_pages, self.s_pages = r_pages, s_pages
        n_jalr, since_link = int(counts.get("Jalr", 0)), 1 << 30
        for idx in order:
            name = kinds[idx][0]
            cls_b = bool(rng.integers(2))
            if name == "Jalr":
                # `auipc x21, 0` every <= 400 instructions keeps a code address in the link register (class A); a JALR then
                # returns to its own pc + 4 through it: target = x21 + imm (the call / return idiom of compiled code)
                if since_link > 400:
                    emit("UType", "AUIPC", LINK_REG, -1, -1, 0, True)
                    since_link = 0
                emit("Jalr", "JALR", pick(D2_REGS), LINK_REG, -1, 4 * (since_link + 1) + 4, True)
                since_link += 1
                continue
            since_link += 1
            if name == "Ecall":
                code = ECALL_CODES[int(rng.integers(len(ECALL_CODES)))]
                lo = code & 0x7FF
                if code >> 11:
                    emit("UType", "LUI", SYSCALL_HI_REG, -1, -1, code - lo, True)
                    emit("Addi", "ADDI", SYSCALL_REG, SYSCALL_HI_REG, -1, lo, True)
                    since_link += 2
                else:
                    emit("Addi", "ADDI", SYSCALL_REG, 0, -1, lo, True)
                    since_link += 1
                emit("SyscallInstrs", "ECALL", SYSCALL_REG, PR_REGS[1], PS_REGS[0], 0, False)
                continue
            if name in ALU_KINDS:
                opn = pick(ALU_KINDS[name])
                srcs = B_REGS + [0] if not cls_b else B_REGS + D1_REGS + PR_REGS + PS_REGS + [0]
                dst = pick(D1_REGS) if not cls_b else pick(D2_REGS)
                use_imm = name == "Addi" or (name in IMM_CAPABLE and bool(rng.integers(2)))
                if use_imm:
                    if name in ("ShiftLeft", "ShiftRight"):
                        v = int(rng.integers(0, 32 if opn.endswith("W") else 64))
                    else:
                        v = int(rng.integers(-2048, 2048))
                    emit(name, opn, dst, pick(srcs), -1, v, True)
                else:
                    emit(name, opn, dst, pick(srcs), pick(srcs), 0, False)
            elif name == "UType":
                which = int(rng.integers(4))
                if which == 0:                                                  # LUI: a pointer into the read-only region
                    emit(name, "LUI", pick(PR_REGS), -1, -1, self.r_base + PAGE * int(rng.integers(r_pages)), True)
                elif which == 1:
                    emit(name, "LUI", pick(PS_REGS), -1, -1, self.s_base + PAGE * int(rng.integers(s_pages)), True)
                else:
                    v = int(rng.integers(-(1 << 19), 1 << 19)) << 12
                    emit(name, "AUIPC" if which == 2 else "LUI", pick(D1_REGS), -1, -1, v, True)
            elif name in LOAD_KINDS:
                opn = pick(LOAD_KINDS[name])
                nb = ACCESS_BYTES[opn]
                emit(name, opn, 0 if name == "LoadX0" else pick(D2_REGS), pick(PR_REGS), -1, int(rng.integers(0, PAGE // nb)) * nb, True)
            elif name in STORE_KINDS:
                opn = pick(STORE_KINDS[name])
                nb = ACCESS_BYTES[opn]
                # op_a = rs2 (the stored register), op_b = rs1 (the pointer), op_c = offset (disassembler/rrs.rs:L37-L39)
                emit(name, opn, pick(B_REGS + D1_REGS + D2_REGS + [0]), pick(PS_REGS), -1, int(rng.integers(0, PAGE // nb)) * nb, True)
            elif name == "Branch":
                # op_a = rs1, op_b = rs2, op_c = offset; a taken branch lands on pc + 4 as well
                regs = B_REGS + D1_REGS + D2_REGS + [0]
                a = pick(regs)
                emit(name, pick(BRANCH_OPS), a, a if rng.integers(4) == 0 else pick(regs), -1, 4, True)
            elif name == "Jal":
                emit(name, "JAL", pick(D1_REGS), -1, -1, 4, True)
        self.steps = [int(rng.integers(-(1 << 40), 1 << 40)) for _ in B_REGS]
        for r, st in zip(B_REGS, self.steps):
            # the tail: scalars advance by a 12-bit immediate (the "step" of iteration k is imm, wide values come from init)
            emit("Addi", "ADDI", r, r, -1, int(st % 4096) - 2048, True)
        L = len(op) + 1
        emit("Jal", "JAL", 0, -1, -1, -4 * (L - 1), True)
        self.L = L
        self.chip = np.array(chip)
        self.op, self.rd, self.rs1, self.rs2 = (np.array(a, dtype=np.int64) for a in (op, rd, rs1, rs2))
        self.imm, self.has_imm = np.array(imm, dtype=np.int64), np.array(has_imm, dtype=bool)
        self.pc_base = pc_base
        self.n_tail = len(B_REGS) + 1
        # clock: 8 per instruction (CLK_INC), an ECALL advances it by 8 + 256
        self.clk_inc = np.where(self.chip == "SyscallInstrs", 8 + ECALL_EXTRA_CLK, 8).astype(np.int64)
        self.clk_off = np.concatenate([[0], np.cumsum(self.clk_inc)[:-1]]).astype(np.int64)
        self.clk_per_iter = int(self.clk_inc.sum())


# which register sits in which access slot, per chip family (None = no access in that slot)
def _slots(body):
    """slot A, B, C register numbers per body position (-1 = no access). Stores / branches read op_a (their rs in slot A)."""
    A = body.rd.copy()
    B = body.rs1.copy()
    C = np.where(body.has_imm, -1, body.rs2)
    is_st_br = np.isin(body.chip, list(STORE_KINDS) + ["Branch"])
    # stores / branches: rd field holds op_a (a READ), rs1 holds op_b
    return A, B, C, is_st_br


class Execution:
    def __init__(self, counts, K=1, seed=0, clk0=1, pc_base=0x200000, mem_pages=(4, 4), device="cpu"):
        rng = np.random.default_rng(seed)
        self.dev = torch.device(device)
        self.body = b = Body(counts, rng, pc_base, mem_pages)
        self.K, self.L, self.clk0 = K, b.L, clk0
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(seed + 1)
        t = lambda a: torch.as_tensor(a, device=self.dev)
        L, dev = b.L, self.dev
        self.op, self.imm, self.has_imm = t(b.op), t(b.imm), t(b.has_imm)
        self.clk_off, self.clk_inc = t(b.clk_off), t(b.clk_inc)
        sA, sB, sC, st_br = _slots(b)
        self.slot_reg = {"A": t(sA), "B": t(sB), "C": t(sC)}
        self.writes_rd = t(~st_br & (sA >= 0))                    # slot A is a write of a new value
        self.init = torch.randint(MIN64, (1 << 63) - 1, (32,), generator=self.gen, device=dev, dtype=I64)
        self.init[0] = 0
        for r in PR_REGS:
            self.init[r] = b.r_base
        for r in PS_REGS:
            self.init[r] = b.s_base
        # ---- static last-writer tables
        pos = np.arange(L)
        writer_pos = {r: pos[(sA == r) & ~st_br] for r in range(32)}
        self.lastw = t(np.array([wp[-1] if len(wp) else -1 for wp in writer_pos.values()], dtype=np.int64))
        lw = {}
        for s, regs in (("A", sA), ("B", sB), ("C", sC)):
            out = np.full(L, -1, dtype=np.int64)
            for r in range(32):
                m = regs == r
                if m.any() and len(writer_pos[r]):
                    i = np.searchsorted(writer_pos[r], pos[m], side="left") - 1        # last writer strictly before p
                    out[m] = np.where(i >= 0, writer_pos[r][np.maximum(i, 0)], -1)
            lw[s] = t(out)
        self.lw = lw
        # ---- static previous-access tables: accesses sorted by (reg, p, time order C < B < A)
        rows = []
        for s, regs, o in (("C", sC, 0), ("B", sB, 1), ("A", sA, 2)):
            m = regs >= 0
            rows.append(np.stack([regs[m], pos[m], np.full(m.sum(), o)], axis=1))
        acc = np.concatenate(rows)
        acc = acc[np.lexsort((acc[:, 2], acc[:, 1], acc[:, 0]))]
        first = np.r_[True, acc[1:, 0] != acc[:-1, 0]]
        last_of_reg = np.r_[first[1:], True]
        prev_idx = np.arange(len(acc)) - 1
        # the first access of a register in an iteration follows the register's LAST access of the previous iteration
        seg_last = np.zeros(len(acc), dtype=np.int64)
        ends = np.nonzero(last_of_reg)[0]
        starts = np.nonzero(first)[0]
        for s0, e0 in zip(starts, ends):
            seg_last[s0:e0 + 1] = e0
        prev_idx = np.where(first, seg_last, prev_idx)
        off = np.array([2, 3, 4])
        self.prev = {}
        for s, o in (("C", 0), ("B", 1), ("A", 2)):
            m = acc[:, 2] == o
            pp, po, wrap = (np.full(L, -1, dtype=np.int64) for _ in range(3))
            pp[acc[m, 1]] = acc[prev_idx[m], 1]
            po[acc[m, 1]] = off[acc[prev_idx[m], 2]]
            wrap[acc[m, 1]] = first[m]
            self.prev[s] = (t(pp), t(po), t(wrap))
        self.touched_regs = sorted(set(int(r) for r in acc[:, 0]))
        self.last_access = {int(acc[e, 0]): (int(acc[e, 1]), int(off[acc[e, 2]])) for e in ends}
        # ---- values: W[k, p] = value written by position p in iteration k
        self.W = torch.zeros((K, L), dtype=I64, device=dev)
        self.kk = torch.arange(K, device=dev)
        self._run()

    # -- time and pc
    def T(self, k, p):
        return self.clk0 + k * self.body.clk_per_iter + self.clk_off[p]

    def pc(self, p):
        return self.body.pc_base + 4 * p

    def reg_value(self, k, p, slot):
        """Value of the register in `slot` of position p just before instruction (k, p). k, p: index tensors of one shape."""
        reg = self.slot_reg[slot][p]
        lw = self.lw[slot][p]
        lastw = self.lastw[reg.clamp(min=0)]
        same = lw >= 0
        prev_it = (~same) & (lastw >= 0) & (k > 0)
        v = torch.where(same, self.W[k, lw.clamp(min=0)],
                        torch.where(prev_it, self.W[(k - 1).clamp(min=0), lastw.clamp(min=0)], self.init[reg.clamp(min=0)]))
        return torch.where(reg == 0, torch.zeros_like(v), v)

    def _grid(self, positions):
        p = torch.as_tensor(positions, device=self.dev)
        k = self.kk[:, None].expand(self.K, len(p)).reshape(-1)
        return k, p[None, :].expand(self.K, len(p)).reshape(-1)

    def _run(self):
        b, L = self.body, self.L
        pos = np.arange(L)
        tail = pos[L - b.n_tail:L - 1]
        # tail ADDIs: W[k, tail_r] = init_r + (k + 1) imm_r
        tp = torch.as_tensor(tail, device=self.dev)
        self.W[:, tp] = self.init[self.slot_reg["A"][tp]][None, :] + (self.kk[:, None] + 1) * self.imm[tp][None, :]
        sB, sC = b.rs1, np.where(b.has_imm, -1, b.rs2)
        is_st_br = np.isin(b.chip, list(STORE_KINDS) + ["Branch"])
        writer = ~is_st_br & (b.rd > 0)
        body_pos = pos < L - b.n_tail
        d1like = set(D1_REGS + PR_REGS + PS_REGS + [LINK_REG, SYSCALL_HI_REG])
        cls_a = writer & body_pos & np.array([int(r) in d1like for r in b.rd])
        cls_c = writer & body_pos & (b.chip == "SyscallInstrs")           # an ECALL re-writes the value its `addi` (class B) produced
        cls_b = writer & body_pos & ~cls_a & ~cls_c
        self.mem = None
        for mask in (cls_a, cls_b, cls_c):
            if not mask.any():
                continue
            k, p = self._grid(pos[mask])
            self.W[k, p] = self._semantics(k, p)

    def operands(self, k, p):
        bval = self.reg_value(k, p, "B")
        cval = torch.where(self.has_imm[p], self.imm[p], self.reg_value(k, p, "C"))
        return bval, cval

    def load_value(self, addr):
        """64-bit word of the read-only region containing byte address `addr` (a fixed pseudo-random image)."""
        w = addr >> 3
        z = w * -7046029254386353131 + 1442695040888963407          # splitmix-style mixing, wrapping int64 arithmetic
        z = (z ^ srl(z, torch.full_like(z, 30))) * -4658895280553007687
        z = (z ^ srl(z, torch.full_like(z, 27))) * -7723592293110705685
        return z ^ srl(z, torch.full_like(z, 31))

    def _semantics(self, k, p):
        op = self.op[p]
        bv, cv = self.operands(k, p)
        pc = self.pc(p)
        out = torch.zeros_like(bv)
        sh6, sh5 = cv & 63, cv & 31

        def put(name, val):
            nonlocal out
            out = torch.where(op == OPC[name], val, out)
        put("ADD", bv + cv); put("ADDI", bv + cv); put("SUB", bv - cv)
        put("XOR", bv ^ cv); put("OR", bv | cv); put("AND", bv & cv)
        put("SLT", (bv < cv).to(I64)); put("SLTU", ult(bv, cv).to(I64))
        put("SLL", bv << sh6); put("SLLW", sext32(bv << sh5))
        put("SRL", srl(bv, sh6)); put("SRA", bv >> sh6)
        put("SRLW", sext32(srl(bv & MASK32, sh5))); put("SRAW", sext32(bv) >> sh5)
        put("ADDW", sext32(bv + cv)); put("SUBW", sext32(bv - cv))
        put("MUL", bv * cv); put("MULW", sext32(bv * cv))
        for name in ("MULH", "MULHU", "MULHSU"):
            m = op == OPC[name]
            if bool(m.any()):
                out = torch.where(m, mulh(bv, cv, name), out)
        m_ec = op == OPC["ECALL"]
        if bool(m_ec.any()):
            out = torch.where(m_ec, self.reg_value(k, p, "A"), out)
        m_div = torch.zeros_like(op, dtype=torch.bool)
        for name in ALU_KINDS["DivRem"]:
            m_div |= op == OPC[name]
        if bool(m_div.any()):
            idx = torch.nonzero(m_div)[:, 0]
            vals = [divrem_result(int(o), int(x), int(y)) for o, x, y in zip(op[idx].tolist(), bv[idx].tolist(), cv[idx].tolist())]
            out[idx] = torch.as_tensor([v[0] for v in vals], dtype=I64, device=out.device)
        put("LUI", self.imm[p]); put("AUIPC", pc + self.imm[p])
        put("JAL", pc + 4); put("JALR", pc + 4)
        isload = torch.zeros_like(op, dtype=torch.bool)
        for name in ("LB", "LBU", "LH", "LHU", "LW", "LWU", "LD"):
            isload |= op == OPC[name]
        if bool(isload.any()):
            addr = bv + cv
            word = self.load_value(addr)
            sh = (addr & 7) * 8
            raw = srl(word, sh)
            put("LD", word)
            put("LWU", raw & MASK32); put("LW", sext32(raw))
            put("LHU", raw & MASK16); put("LH", (raw & MASK16) - (((raw >> 15) & 1) << 16))
            put("LBU", raw & 0xFF); put("LB", (raw & 0xFF) - (((raw >> 7) & 1) << 8))
        return out


U64 = (1 << 64) - 1
_S64 = lambda v: v - (1 << 64) if v >> 63 else v                      # u64 -> i64 (python ints)
_S32 = lambda v: (v & MASK32) - (1 << 32) if (v >> 31) & 1 else v & MASK32
_TDIV = lambda a, b: abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)      # truncating division


def divrem_result(op, b, c):
    """(value written to rd, quotient, remainder), all as SIGNED int64 python ints; b, c signed int64.
    `get_quotient_and_remainder` (core/executor/src/utils.rs:L73-L93) + the opcode's choice."""
    name = OPC_NAME[op]
    bu, cu = b & U64, c & U64
    word = name.endswith("W")
    signed = name in ("DIV", "REM", "DIVW", "REMW")
    if not word and cu == 0:
        q, r = U64, bu
    elif word and (cu & MASK32) == 0:
        q, r = U64, _S32(bu) & U64
    elif signed and not word:
        bs, cs = _S64(bu), _S64(cu)
        q = _TDIV(bs, cs)
        q, r = q & U64, (bs - _TDIV(bs, cs) * cs) & U64
    elif signed and word:
        bs, cs = _S32(bu), _S32(cu)
        q = _TDIV(bs, cs)
        r = bs - q * cs
        q, r = _S32(q & MASK32) & U64, _S32(r & MASK32) & U64
    elif word:
        bb, cc = bu & MASK32, cu & MASK32
        q, r = _S32(bb // cc) & U64, _S32(bb % cc) & U64
    else:
        q, r = bu // cu, bu % cu
    a = q if name.startswith("DIV") else r
    return _S64(a), q, r


OPC_NAME = {v: k for k, v in OPC.items()}


def divrem_rows(L, width, n_padded, ops, bvals, cvals):
    """DivRem's own columns (everything but the CPU state and the register-access bookkeeping of the adapter) for the events
    (opcode number, b, c) — b, c signed int64 python ints — plus the reference's padding rows: alu/divrem/mod.rs:L262-L586
    (event_to_row; padding rows = 0 / 1). Returns canonical [n_padded, width] int64."""
    rows = np.zeros((n_padded, width), dtype=np.int64)
    n = len(ops)

    def put(r, name, vals, off=0):
        vals = vals if isinstance(vals, (list, tuple)) else [vals]
        c0 = L[name] + off
        rows[r, c0:c0 + len(vals)] = vals
    w16 = lambda v: [(v >> (16 * i)) & MASK16 for i in range(4)]
    inv = lambda v: pow(v % P, P - 2, P) if v % P else 0

    def is_zero_word(r, prefix, limbs):                    # IsZeroWordOperation::populate_from_field_element
        res = []
        for i, x in enumerate(limbs):
            put(r, "%s.is_zero_limb.%d.inverse" % (prefix, i), inv(x))
            put(r, "%s.is_zero_limb.%d.result" % (prefix, i), int(x % P == 0))
            res.append(int(x % P == 0))
        put(r, prefix + ".is_zero_first_half", res[0] * res[1])
        put(r, prefix + ".is_zero_second_half", res[2] * res[3])
        put(r, prefix + ".result", int(all(res)))
        return int(all(res))

    def mul_op(r, prefix, x, y, is_mulh):                   # MulOperation::populate (operations/mul.rs:L54-L137)
        xb = [(x >> (8 * i)) & 0xFF for i in range(8)]
        yb = [(y >> (8 * i)) & 0xFF for i in range(8)]
        b_msb, c_msb = xb[7] >> 7, yb[7] >> 7
        bse, cse = (b_msb if is_mulh else 0), (c_msb if is_mulh else 0)
        xe, ye = xb + [bse * 0xFF] * 8, yb + [cse * 0xFF] * 8
        prod = [0] * 16
        for i in range(16):
            for j in range(16 - i):
                prod[i + j] += xe[i] * ye[j]
        carry = [0] * 16
        for i in range(16):
            carry[i] = prod[i] >> 8
            prod[i] &= 0xFF
            if i + 1 < 16:
                prod[i + 1] += carry[i]
        put(r, prefix + ".carry", carry)
        put(r, prefix + ".product", prod)
        put(r, prefix + ".b_lower_byte.low_bytes", [v & 0xFF for v in w16(x)])
        put(r, prefix + ".c_lower_byte.low_bytes", [v & 0xFF for v in w16(y)])
        put(r, prefix + ".b_msb", b_msb)
        put(r, prefix + ".c_msb", c_msb)
        put(r, prefix + ".b_sign_extend", bse)
        put(r, prefix + ".c_sign_extend", cse)

    for r, (op, b_, c_) in enumerate(zip(ops, bvals, cvals)):
        name = OPC_NAME[op]
        word, signed = name.endswith("W"), name in ("DIV", "REM", "DIVW", "REMW")
        eb, ec = b_ & U64, c_ & U64                                     # event.b / event.c
        a_, q, rem = divrem_result(op, b_, c_)
        bq = (_S32(eb) & U64) if word and signed else (eb & MASK32) if word else eb
        cq = (_S32(ec) & U64) if word and signed else (ec & MASK32) if word else ec
        put(r, "a", w16(a_ & U64)); put(r, "b", w16(bq)); put(r, "c", w16(cq))
        put(r, "is_real", 1)
        put(r, "is_" + name.lower(), 1)
        put(r, "is_real_not_word", int(not word))
        c0 = is_zero_word(r, "is_c_0", w16(cq))
        put(r, "quotient", w16(q)); put(r, "remainder", w16(rem))
        qc = (q & MASK32) if word and not signed else q
        rc = (rem & MASK32) if word and not signed else rem
        put(r, "quotient_comp", w16(qc)); put(r, "remainder_comp", w16(rc))
        rem_neg = b_neg = c_neg = overflow = 0
        if signed and not word:
            rem_neg, b_neg, c_neg = rem >> 63, eb >> 63, ec >> 63
            overflow = int(eb == 1 << 63 and ec == U64)
            abs_rem, abs_c = abs(_S64(rem)), abs(_S64(ec))
        elif signed:
            rem_neg, b_neg, c_neg = (rem >> 31) & 1, (eb >> 31) & 1, (ec >> 31) & 1
            overflow = int(eb & MASK32 == 1 << 31 and ec & MASK32 == MASK32)
            abs_rem, abs_c = abs(_S64(rem)), abs(_S64(cq))
        elif word:
            abs_rem, abs_c = rc, ec & MASK32
        else:
            abs_rem, abs_c = rc, ec
        put(r, "rem_neg", rem_neg); put(r, "b_neg", b_neg); put(r, "c_neg", c_neg); put(r, "is_overflow", overflow)
        put(r, "abs_remainder", w16(abs_rem)); put(r, "abs_c", w16(abs_c)); put(r, "max_abs_c_or_1", w16(max(1, abs_c)))
        ob, oc = ((eb & MASK32, 1 << 31), (ec & MASK32, MASK32)) if word else ((eb, 1 << 63), (ec, U64))
        is_zero_word(r, "is_overflow_b", [x - y for x, y in zip(w16(ob[0]), w16(ob[1]))])
        is_zero_word(r, "is_overflow_c", [x - y for x, y in zip(w16(oc[0]), w16(oc[1]))])
        put(r, "b_neg_not_overflow", b_neg * (1 - overflow))
        put(r, "b_not_neg_not_overflow", (1 - b_neg) * (1 - overflow))
        put(r, "abs_c_alu_event", c_neg); put(r, "abs_rem_alu_event", rem_neg)
        if c_neg:
            put(r, "c_neg_operation.value", w16((cq + abs_c) & U64))
        if rem_neg:
            put(r, "rem_neg_operation.value", w16((rem + abs_rem) & U64))
        if word:
            put(r, "b_msb", (eb >> 31) & 1); put(r, "c_msb", (ec >> 31) & 1)
            put(r, "rem_msb", (rem >> 31) & 1); put(r, "quot_msb", (q >> 31) & 1)
        else:
            put(r, "b_msb", bq >> 63); put(r, "c_msb", cq >> 63); put(r, "rem_msb", rem >> 63)
        put(r, "remainder_check_multiplicity", 1 - c0)
        if not c0:                                            # LtOperationUnsigned::populate_unsigned(abs_rem, max(abs_c, 1))
            xl, yl = w16(abs_rem), w16(max(1, abs_c))
            d = [i for i in (3, 2, 1, 0) if xl[i] != yl[i]]
            if d:
                i = d[0]
                flags = [int(j == i) for j in range(4)]
                put(r, "remainder_lt_operation.u16_flags", flags)
                put(r, "remainder_lt_operation.comparison_limbs", [xl[i], yl[i]])
                put(r, "remainder_lt_operation.not_eq_inv", inv(xl[i] - yl[i]))
                put(r, "remainder_lt_operation.bit", int(xl[i] < yl[i]))
        lower = (qc * cq) & U64
        if signed:
            upper = ((_S64(qc) * _S64(cq)) >> 64) & U64
        else:
            upper = ((qc * cq) >> 64) & U64
        ctq = w16(lower) + w16(upper)
        put(r, "c_times_quotient", ctq)
        mul_op(r, "c_times_quotient_lower", qc, cq, False)
        if not word:
            mul_op(r, "c_times_quotient_upper", qc, cq, signed)
        rem16 = w16(rc) + [rem_neg * MASK16] * 4
        carry, prev = [], 0
        for i in range(8):
            x = ctq[i] + rem16[i] + prev
            prev = x >> 16
            carry.append(prev)
        put(r, "carry", carry)
    for r in range(n, rows.shape[0]):                         # padding rows: 0 / 1 (quotient = remainder = 0)
        put(r, "is_divu", 1)
        put(r, "adapter.op_c_memory.prev_value", [1, 0, 0, 0])
        put(r, "abs_c", [1, 0, 0, 0]); put(r, "c", [1, 0, 0, 0]); put(r, "max_abs_c_or_1", [1, 0, 0, 0])
        put(r, "b_not_neg_not_overflow", 1)
        is_zero_word(r, "is_c_0", [1, 0, 0, 0])
    for r, (b_, c_) in enumerate(zip(bvals, cvals)):              # the operands as the adapter's register reads see them
        put(r, "adapter.op_b_memory.prev_value", w16(b_ & U64))
        put(r, "adapter.op_c_memory.prev_value", w16(c_ & U64))
    return rows % P




def mulh(b, c, name):
    """High 64 bits of the 128-bit product (signedness per opcode) through 16-bit limbs."""
    bs = name in ("MULH", "MULHSU")
    cs = name == "MULH"
    bl = [(b >> (16 * i)) & MASK16 for i in range(4)] + [((b >> 63) & 1) * MASK16 * int(bs)] * 4
    cl = [(c >> (16 * i)) & MASK16 for i in range(4)] + [((c >> 63) & 1) * MASK16 * int(cs)] * 4
    acc = [torch.zeros_like(b) for _ in range(8)]
    for i in range(8):
        for j in range(8 - i):
            acc[i + j] = acc[i + j] + bl[i] * cl[j]
    carry = torch.zeros_like(b)
    limbs = []
    for i in range(8):
        s = acc[i] + carry
        limbs.append(s & MASK16)
        carry = s >> 16
    return limbs[4] | (limbs[5] << 16) | (limbs[6] << 32) | (limbs[7] << 48)


# ---------------------------------------------------------------------------------------------------------------------
class Table:
    def __init__(self, air, n, dev):
        self.air, self.n = air, n
        self.main = torch.zeros((pad32(n), air.main_width), dtype=I64, device=dev)
        self.prep = torch.zeros((pad32(n), air.prep_width), dtype=I64, device=dev) if air.prep_width else None
        self.L = getattr(air, "layout", {})

    def set(self, name, val, off=0):
        c = self.L[name] + off
        if not torch.is_tensor(val):
            self.main[:self.n, c] = val
        elif val.dim() == 1:
            self.main[:self.n, c] = val
        else:
            self.main[:self.n, c:c + val.shape[1]] = val


class EmptyTable(Table):
    """A chip of the cluster without events: no rows at all (not even padding rows)."""

    def __init__(self, air, dev):
        self.air, self.n = air, 0
        self.main = torch.zeros((0, air.main_width), dtype=I64, device=dev)
        self.prep = None
        self.L = getattr(air, "layout", {})


# the shape clusters of RiscvAir::machine (core/machine/src/riscv/mod.rs:L547-L803, `mprotect` off), by chip name
_PREPROCESSED = ["Program", "Byte", "Range"]
CORE_CLUSTER = _PREPROCESSED + ["SyscallCore", "DivRem", "Add", "Addi", "Addw", "Sub", "Subw", "Bitwise", "Mul", "ShiftRight", "ShiftLeft", "Lt", "AluX0",
                                "LoadByte", "LoadHalf", "LoadWord", "LoadDouble", "LoadX0", "StoreByte", "StoreHalf", "StoreWord", "StoreDouble",
                                "UType", "Branch", "Jal", "Jalr", "SyscallInstrs", "MemoryBump", "StateBump", "MemoryLocal", "Global"]
MEMORY_CLUSTER = _PREPROCESSED + ["MemoryGlobalInit", "MemoryGlobalFinalize", "Global"]
_PRECOMPILE_BASE = _PREPROCESSED + ["SyscallPrecompile", "MemoryLocal", "Global"]
PRECOMPILE_CLUSTERS = [_PRECOMPILE_BASE + c for c in (
    ["ShaExtend", "ShaExtendControl"], ["ShaCompress", "ShaCompressControl"], ["EdAddAssign"], ["EdDecompress"], ["Secp256k1AddAssign"],
    ["Secp256k1DoubleAssign"], ["Secp256r1AddAssign"], ["Secp256r1DoubleAssign"], ["KeccakPermute", "KeccakPermuteControl"], ["Bn254AddAssign"],
    ["Bn254DoubleAssign"], ["Bls12381AddAssign"], ["Bls12381DoubleAssign"], ["Uint256MulMod"], ["Uint256Ops"], ["Bls12381FpOpAssign"],
    ["Bls12381Fp2AddSubAssign"], ["Bls12381Fp2MulAssign"], ["Bn254FpOpAssign"], ["Bn254Fp2AddSubAssign"], ["Bn254Fp2MulAssign"], ["Poseidon2"])]
_CORE_EXTENSIONS = [["MemoryGlobalInit", "MemoryGlobalFinalize"], ["Bls12381FpOpAssign"], ["Bn254FpOpAssign"],
                    ["ShaExtend", "ShaExtendControl", "ShaCompress", "ShaCompressControl"], ["Uint256Ops"], ["Poseidon2"]]


def chip_clusters():
    """`machine.shape().chip_clusters` as frozensets of chip names: the core cluster alone, with one extension, with all six, the
    special one, the memory cluster, the 22 precompile clusters (riscv/mod.rs:L717-L790). A shard's chip set must BE one of them
    (`ShardVerifier::verify_shard`, verifier/shard.rs:L537)."""
    core = [CORE_CLUSTER] + [CORE_CLUSTER + e for e in _CORE_EXTENSIONS] + [CORE_CLUSTER + [n for e in _CORE_EXTENSIONS for n in e]]
    core.append(CORE_CLUSTER + ["MemoryGlobalInit", "MemoryGlobalFinalize", "ShaExtend", "ShaExtendControl", "ShaCompress", "ShaCompressControl", "Uint256Ops"])
    return [frozenset(c) for c in core + [MEMORY_CLUSTER] + PRECOMPILE_CLUSTERS]


def smallest_cluster(names):
    """`MachineShape::smallest_cluster` (hypercube/src/machine.rs:L29-L35)."""
    fits = [c for c in chip_clusters() if set(names) <= c]
    assert fits, ("no shape cluster holds", sorted(names))
    return min(fits, key=len)


def program_table(program, pc_base, dev, executed_pc=None):
    """Program: one row per instruction of the text — `program` [n, 6] = (opcode, op_a, op_b, op_c, imm_b, imm_c), the
    transpiled instruction list — (program/trusted.rs:L80-L131); multiplicity = how often each pc of `executed_pc` occurs (none:
    a shard that executes nothing — every cluster holds the Program chip, riscv/mod.rs:L547-L560)."""
    prog = torch.as_tensor(program, device=dev)
    n = prog.shape[0]
    air, it = R.chip("Program")
    tb = Table(air, n, dev)
    pc = pc_base + 4 * torch.arange(n, device=dev)
    op, a, b_, c_, imm_b, imm_c = (prog[:, i] for i in range(6))
    tb.prep[:n, 0:3] = limbs16(pc)[:, :3]
    tb.prep[:n, 3], tb.prep[:n, 4] = op, a
    regw = lambda r: torch.stack([r] + [torch.zeros_like(r)] * 3, dim=1)
    tb.prep[:n, 5:9] = torch.where((imm_b == 1)[:, None], limbs16(b_), regw(b_))
    tb.prep[:n, 9:13] = torch.where((imm_c == 1)[:, None], limbs16(c_), regw(c_))
    tb.prep[:n, 13] = (a == 0).to(I64)
    tb.prep[:n, 14], tb.prep[:n, 15] = imm_b, imm_c
    if executed_pc is not None:
        tb.main[:n, 0] = torch.bincount((executed_pc - pc_base) >> 2, minlength=n)
    if tb.prep.shape[0] > n:
        tb.prep[n:] = tb.prep[0]
    return tb, (air, it)


class RunContext:
    """What a shard that executes no instructions still takes from the run it belongs to: the program (its Program table, all
    multiplicities zero, and the entry point a precompile shard's public values name: `update_initialized_state`,
    public_values.rs:L272-L299) and, for the memory shards, the final state of the execution (`update_finalized_state`).
    The default is a one-instruction program that has run to HALT without committing anything: what the stand-alone shard
    builders of riscv_more_trace.py (tests, `bench.py --workload precompile`) are set in."""

    def __init__(self, program=None, pc_base=0x200000, pc_start=None, final=None):
        self.program = np.array([[OPC["ADD"], 0, 0, 0, 0, 0]], dtype=np.int64) if program is None else program
        self.pc_base, self.pc_start = pc_base, pc_base if pc_start is None else pc_start
        # (timestamp, pc, exit code, committed_value_digest[8 words], deferred_proofs_digest[8])
        self.final = final if final is not None else (9, 1, 0, [0] * 8, [0] * 8)

    def program_table(self, dev):
        return program_table(self.program, self.pc_base, dev)


def generate(counts, K=1, seed=0, clk0=1, pc_base=0x200000, mem_pages=(4, 4), device="cpu", real_global=True):
    """counts: {instruction kind: positions in the loop body} over Add, Addi, Sub, Bitwise, Lt, Mul, ShiftLeft, ShiftRight,
    Addw, Subw, UType, LoadByte/Half/Word/Double, StoreByte/Half/Word/Double, Branch, Jal, Jalr. The body is executed K times.
    Returns (machine, tables, public_values): machine = [(AirProgram, InteractionProgram)] in chip-name order,
    tables = {name: (prep, main)} canonical int64 [rows, width] tensors on `device` (rows padded to multiples of 32)."""
    ex = Execution(counts, K, seed, clk0, pc_base, mem_pages, device)
    tr = Tracer(ex)
    tr.real_global = real_global
    return tr.build()


class Tracer:
    def __init__(self, ex):
        self.ex, self.dev, self.b = ex, ex.dev, ex.body
        self.tables = {}
        self.bumps = []              # MemoryBump events: (reg, prev_ts, value, ts)
        self.state_bumps = []        # (clk_high, clk_low_sent, pc_sent[3], next pc value, next clk value)

    # -- shared column groups
    def fill_state(self, tb, k, p):
        T = self.ex.T(k, p)
        tb.set("state.clk_high", T >> 24)
        tb.set("state.clk_16_24", (T >> 16) & 0xFF)
        tb.set("state.clk_0_16", T & MASK16)
        pc = self.ex.pc(p)
        tb.set("state.pc", torch.stack([(pc >> (16 * i)) & MASK16 for i in range(3)], dim=1))

    def fill_reg_access(self, tb, prefix, k, p, slot, live=None):
        """RegisterAccessCols for the access in `slot` (memory/consistency/trace.rs:L22-L33, L104-L127). Returns prev value."""
        ex = self.ex
        val = ex.reg_value(k, p, slot)
        t_prev = self._prev_ts(k, p, slot)
        t_cur = ex.T(k, p) + POS_OFF[slot]
        cross = (t_prev >> 24) != (t_cur >> 24)
        if live is not None:
            cross = cross & live
        if bool(cross.any()):
            reg = ex.slot_reg[slot][p]
            self.bumps.append((reg[cross], t_prev[cross], val[cross], (t_cur[cross] >> 24) << 24))
        old = torch.where(cross, torch.zeros_like(t_prev), t_prev & 0xFFFFFF)
        diff = (t_cur & 0xFFFFFF) - old - 1
        if live is not None:                                   # an immediate operand: no access, all-zero timestamp columns
            old, diff = old * live, diff * live
        tb.set(prefix + ".prev_value", limbs16(val))
        tb.set(prefix + ".prev_low", old)
        tb.set(prefix + ".diff_low_limb", diff & MASK16)
        return val

    def _prev_ts(self, k, p, slot):
        """Timestamp of the previous access of the register in `slot` (0 = the initial record)."""
        ex = self.ex
        pp, po, wrap = (x[p] for x in ex.prev[slot])
        kp = k - wrap
        return torch.where(kp >= 0, ex.T(kp.clamp(min=0), pp) + po, torch.zeros_like(k))

    def rows_of(self, chip):
        pos = np.nonzero(self.b.chip == chip)[0]
        k, p = self.ex._grid(pos)
        return k, p

    def table(self, chip, n):
        air, _ = R.chip(chip)
        tb = Table(air, n, self.dev)
        self.tables[chip] = tb
        return tb

    # -- chips with the R / I / ALU adapters
    def fill_adapter(self, tb, k, p, kind):
        ex = self.ex
        A, Bq = ex.slot_reg["A"][p], ex.slot_reg["B"][p]
        tb.set("adapter.op_a", A.clamp(min=0))
        tb.set("adapter.op_a_0", (A == 0).to(I64))
        a_prev = self.fill_reg_access(tb, "adapter.op_a_memory", k, p, "A")
        bval = cval = None
        if kind in ("R", "I", "ALU"):
            tb.set("adapter.op_b", Bq.clamp(min=0))
            bval = self.fill_reg_access(tb, "adapter.op_b_memory", k, p, "B")
        if kind == "R":
            tb.set("adapter.op_c", ex.slot_reg["C"][p].clamp(min=0))
            cval = self.fill_reg_access(tb, "adapter.op_c_memory", k, p, "C")
        elif kind == "I":
            cval = ex.imm[p]
            tb.set("adapter.op_c_imm", limbs16(cval))
        elif kind == "ALU":
            is_imm = ex.has_imm[p]
            live = ~is_imm
            creg = ex.slot_reg["C"][p]
            cv_reg = self.fill_reg_access(tb, "adapter.op_c_memory", k, p, "C", live=live)
            cval = torch.where(is_imm, ex.imm[p], cv_reg)
            tb.set("adapter.op_c_memory.prev_value", limbs16(cval))
            opc = torch.where(is_imm[:, None], limbs16(ex.imm[p]), torch.stack([creg.clamp(min=0)] + [torch.zeros_like(creg)] * 3, dim=1))
            tb.set("adapter.op_c", opc)
            tb.set("adapter.imm_c", is_imm.to(I64))
        elif kind == "J":
            pass
        return a_prev, bval, cval

    def simple_alu(self, chip, kind, extra):
        k, p = self.rows_of(chip)
        n = len(p)
        if n == 0:
            return
        tb = self.table(chip, n)
        self.fill_state(tb, k, p)
        _, bv, cv = self.fill_adapter(tb, k, p, kind)
        a = self.ex.W[k, p]
        extra(tb, k, p, a, bv, cv)

    def build(self):
        ex, b = self.ex, self.b
        chips_present = set(b.chip.tolist())

        def value_chip(name, kind, flag=None):
            def extra(tb, k, p, a, bv, cv):
                tb.set("value", limbs16(a))
                tb.set("is_real", 1)
            self.simple_alu(name, kind, extra)
        value_chip("Add", "R")
        value_chip("Addi", "I")
        value_chip("Sub", "R")

        def addw_extra(tb, k, p, a, bv, cv):
            la = limbs16(a)
            tb.set("value", la[:, :2])
            tb.set("msb", la[:, 1] >> 15)
            tb.set("is_real", 1)
        self.simple_alu("Addw", "ALU", addw_extra)
        self.simple_alu("Subw", "R", addw_extra)

        def bitwise_extra(tb, k, p, a, bv, cv):
            op = ex.op[p]
            tb.set("b_low_bytes.low_bytes", limbs16(bv) & 0xFF)
            tb.set("c_low_bytes.low_bytes", limbs16(cv) & 0xFF)
            tb.set("result", bytes8(a))
            for nm in ("XOR", "OR", "AND"):
                tb.set("is_" + nm.lower(), (op == OPC[nm]).to(I64))
        self.simple_alu("Bitwise", "ALU", bitwise_extra)

        def lt_extra(tb, k, p, a, bv, cv):
            signed = ex.op[p] == OPC["SLT"]
            tb.set("is_slt", signed.to(I64))
            tb.set("is_sltu", (~signed).to(I64))
            self.fill_lt(tb, "lt", bv, cv, signed)
        self.simple_alu("Lt", "ALU", lt_extra)
        self.simple_alu("Mul", "R", self.mul_extra)
        self.simple_alu("DivRem", "R", self.divrem_extra)
        self.syscall_instrs()
        self.simple_alu("ShiftLeft", "ALU", self.sll_extra)
        self.simple_alu("ShiftRight", "ALU", self.sr_extra)
        self.utype()
        self.jal()
        self.jalr()
        self.branch()
        self.memory_instructions()
        self.state_chain()
        self.memory_local_and_bumps()
        return self.finish()

    def fill_lt(self, tb, prefix, bv, cv, signed):
        """LtOperationSigned::populate_signed / LtOperationUnsigned::populate_unsigned (operations/slt.rs:L50-L174)."""
        bl, cl = limbs16(bv), limbs16(cv)
        s = signed.to(I64)
        tb.set(prefix + ".b_msb", (bl[:, 3] >> 15) * s)
        tb.set(prefix + ".c_msb", (cl[:, 3] >> 15) * s)
        bc, cc = bl.clone(), cl.clone()
        bc[:, 3] ^= s << 15
        cc[:, 3] ^= s << 15
        ne = bc != cc
        # the most significant differing limb
        idx = torch.where(ne[:, 3], 3, torch.where(ne[:, 2], 2, torch.where(ne[:, 1], 1, 0)))
        any_ne = ne.any(dim=1)
        flags = torch.zeros_like(bl)
        flags[torch.arange(len(idx), device=idx.device), idx] = 1
        flags = flags * any_ne[:, None].to(I64)
        bsel = (bc * flags).sum(dim=1)
        csel = (cc * flags).sum(dim=1)
        tb.set(prefix + ".result.u16_flags", flags)
        tb.set(prefix + ".result.comparison_limbs", torch.stack([bsel, csel], dim=1))
        tb.set(prefix + ".result.not_eq_inv", torch.where(any_ne, finv(bsel - csel), torch.zeros_like(bsel)))
        tb.set(prefix + ".result.bit", (bsel < csel).to(I64))

    def mul_extra(self, tb, k, p, a, bv, cv):
        """MulOperation::populate (operations/mul.rs:L54-L137)."""
        op = self.ex.op[p]
        is_ = {nm: op == OPC[nm] for nm in ("MUL", "MULH", "MULHU", "MULHSU", "MULW")}
        for nm, m in is_.items():
            tb.set("is_" + nm.lower(), m.to(I64))
        tb.set("a", limbs16(a))
        bb, cb = bytes8(bv), bytes8(cv)
        b_msb, c_msb = bb[:, 7] >> 7, cb[:, 7] >> 7
        bse = ((is_["MULH"] | is_["MULHSU"]).to(I64)) * b_msb
        cse = is_["MULH"].to(I64) * c_msb
        be = torch.cat([bb, (bse * 0xFF)[:, None].expand(-1, 8)], dim=1)
        ce = torch.cat([cb, (cse * 0xFF)[:, None].expand(-1, 8)], dim=1)
        prod = torch.zeros((len(p), 16), dtype=I64, device=self.dev)
        for i in range(16):
            for j in range(16 - i):
                prod[:, i + j] += be[:, i] * ce[:, j]
        carry = torch.zeros_like(prod)
        for i in range(16):
            carry[:, i] = prod[:, i] >> 8
            prod[:, i] &= 0xFF
            if i + 1 < 16:
                prod[:, i + 1] += carry[:, i]
        tb.set("mul.carry", carry)
        tb.set("mul.product", prod)
        tb.set("mul.b_lower_byte.low_bytes", limbs16(bv) & 0xFF)
        tb.set("mul.c_lower_byte.low_bytes", limbs16(cv) & 0xFF)
        tb.set("mul.b_msb", b_msb)
        tb.set("mul.c_msb", c_msb)
        tb.set("mul.product_msb", is_["MULW"].to(I64) * (limbs16(a)[:, 1] >> 15))
        tb.set("mul.b_sign_extend", bse)
        tb.set("mul.c_sign_extend", cse)

    # -- DivRem (alu/divrem/mod.rs:L262-L563 event_to_row; padding rows L565-L586)
    def divrem_extra(self, tb, k, p, a, bv, cv):
        n, L = len(p), tb.L
        rows = divrem_rows(L, tb.air.main_width, tb.main.shape[0], self.ex.op[p].tolist(), bv.tolist(), cv.tolist())
        keep = tb.main.clone()                                    # state / adapter columns were filled by simple_alu
        tb.main[:] = torch.as_tensor(rows, device=self.dev)
        lo, hi = L["state.clk_high"], L["a"]
        tb.main[:n, lo:hi] = keep[:n, lo:hi]

    # -- SyscallInstrs + SyscallCore (syscall/instructions/trace.rs event_to_row; syscall/chip.rs generate_trace_into)
    def syscall_instrs(self):
        k, p = self.rows_of("SyscallInstrs")
        n = len(p)
        if n == 0:
            return
        ex = self.ex
        tb = self.table("SyscallInstrs", n)
        self.fill_state(tb, k, p)
        code, bv, cv = self.fill_adapter(tb, k, p, "R")
        sid = code & 0xFF
        halt = sid == 0x00
        # next_pc is NOT normalised: limb 0 = pc[0] + 4 (air.rs:L128-L140); HALT jumps to HALT_PC = 1
        pcl = limbs16(ex.pc(p))[:, :3].clone()
        pcl[:, 0] += 4
        pcl = torch.where(halt[:, None], torch.tensor([1, 0, 0], dtype=I64, device=self.dev)[None, :], pcl)
        tb.set("next_pc", pcl)
        tb.set("is_halt", halt.to(I64))
        tb.set("op_a_value", limbs16(ex.W[k, p]))
        tb.set("a_low_bytes.low_bytes", limbs16(code) & 0xFF)
        for nm, c_ in (("is_enter_unconstrained", 0x03), ("is_hint_len", 0xF0), ("is_halt_check", 0x00), ("is_commit", 0x10),
                       ("is_commit_deferred_proofs", 0x1A)):
            d = (sid - c_) % P
            tb.set(nm + ".inverse", torch.where(d == 0, torch.zeros_like(d), finv(d)))
            tb.set(nm + ".result", (d == 0).to(I64))
        # COMMIT / COMMIT_DEFERRED_PROOFS: op_b indexes the digest word, COMMIT's op_c is the word itself (trace.rs:L200-L229)
        commit, cdp = sid == 0x10, sid == 0x1A
        idx = (bv & 7) * (commit | cdp).to(I64)
        bitmap = torch.zeros((n, 8), dtype=I64, device=self.dev)
        bitmap[torch.arange(n, device=self.dev), idx] = (commit | cdp).to(I64)
        tb.set("index_bitmap", bitmap)
        tb.set("expected_public_values_digest", bytes8(cv)[:, :4] * commit.to(I64)[:, None])
        top = (P - 1) >> 16
        tb.set("op_b_range_check", (halt & (limbs16(bv)[:, 1] < top)).to(I64))
        tb.set("op_c_range_check", (cdp & (limbs16(cv)[:, 1] < top)).to(I64))
        tb.set("is_real", 1)
        # SyscallCore: one row per ecall with its own table (syscall/chip.rs: events with should_send)
        send = ((code >> 8) & 0xFF) == 1
        if bool(send.any()):
            T = ex.T(k, p)[send]
            sc = self.table("SyscallCore", int(send.sum()))
            sc.set("clk_high", T >> 24)
            sc.set("clk_low", T & 0xFFFFFF)
            sc.set("syscall_id", sid[send])
            sc.set("arg1", limbs16(bv[send])[:, :3])
            sc.set("arg2", limbs16(cv[send])[:, :3])
            sc.set("is_real", 1)

    def _shift_fields(self, tb, cv, word_op):
        c = cv & MASK16
        tb.set("c_bits", torch.stack([(c >> i) & 1 for i in range(6)], dim=1))
        amount = ((c >> 4) & 1) + 2 * ((c >> 5) & 1) * (~word_op).to(I64)
        sh = torch.zeros((len(c), 4), dtype=I64, device=self.dev)
        sh[torch.arange(len(c), device=self.dev), amount] = 1
        tb.set("shift_u16", sh)
        return c & 15

    def sll_extra(self, tb, k, p, a, bv, cv):
        """ShiftLeftChip::event_to_row (alu/sll/mod.rs)."""
        op = self.ex.op[p]
        w = op == OPC["SLLW"]
        tb.set("is_sll", (~w).to(I64))
        tb.set("is_sllw", w.to(I64))
        tb.set("is_sllw_imm", (w & self.ex.has_imm[p]).to(I64))
        tb.set("a", limbs16(a))
        s = self._shift_fields(tb, cv, w)
        tb.set("v_01", 1 << (s & 3))
        tb.set("v_012", 1 << (s & 7))
        tb.set("v_0123", 1 << s)
        bl = limbs16(bv)
        lower = bl & ((1 << (16 - s))[:, None] - 1)
        higher = bl >> (16 - s)[:, None]
        tb.set("lower_limb", lower)
        tb.set("higher_limb", higher)
        res = lower << s[:, None]
        res[:, 1:] += higher[:, :3]
        tb.set("limb_result", res)
        tb.set("sllw_msb", w.to(I64) * (limbs16(a)[:, 1] >> 15))
        for nm in ("v_01", "v_012", "v_0123"):                     # the padded row template (alu/sll/mod.rs:L154-L160)
            tb.main[tb.n:, tb.L[nm]] = 1

    def sr_extra(self, tb, k, p, a, bv, cv):
        """ShiftRightChip::event_to_row (alu/sr/mod.rs:L239-L312)."""
        op = self.ex.op[p]
        is_ = {nm: op == OPC[nm] for nm in ("SRL", "SRA", "SRLW", "SRAW")}
        for nm, m in is_.items():
            tb.set("is_" + nm.lower(), m.to(I64))
        w = is_["SRLW"] | is_["SRAW"]
        tb.set("is_w_imm", (w & self.ex.has_imm[p]).to(I64))
        tb.set("a", limbs16(a))
        s = self._shift_fields(tb, cv, w)
        tb.set("v_01", 1 << (4 - (s & 3)))
        tb.set("v_012", 1 << (8 - (s & 7)))
        v = 1 << (16 - s)
        tb.set("v_0123", v)
        bl = limbs16(bv)
        msb = torch.where(is_["SRA"], bl[:, 3] >> 15, torch.where(is_["SRAW"], bl[:, 1] >> 15, torch.zeros_like(s)))
        tb.set("b_msb", msb)
        tb.set("sra_msb_v0123", msb * v)
        bl = bl.clone()
        bl[:, 2:] *= (~w).to(I64)[:, None]
        tb.set("srw_msb", w.to(I64) * (limbs16(a)[:, 1] >> 15))
        lower = bl & ((1 << s)[:, None] - 1)
        higher = bl >> s[:, None]
        tb.set("lower_limb", lower)
        tb.set("higher_limb", higher)
        res = higher.clone()
        res[:, :3] += lower[:, 1:] * v[:, None]
        tb.set("limb_result", res)
        for nm, val in (("v_01", 16), ("v_012", 256), ("v_0123", 65536)):      # padded row template (alu/sr/mod.rs:L165-L171)
            tb.main[tb.n:, tb.L[nm]] = val

    def utype(self):
        k, p = self.rows_of("UType")
        if len(p) == 0:
            return
        tb = self.table("UType", len(p))
        self.fill_state(tb, k, p)
        self.fill_adapter(tb, k, p, "J")
        imm = self.ex.imm[p]
        tb.set("adapter.op_b_imm", limbs16(imm))
        tb.set("adapter.op_c_imm", limbs16(imm))
        auipc = self.ex.op[p] == OPC["AUIPC"]
        pc = self.ex.pc(p)
        tb.set("addend", limbs16(pc)[:, :3] * auipc.to(I64)[:, None])
        tb.set("value", limbs16(self.ex.W[k, p]))
        tb.set("is_auipc", auipc.to(I64))
        tb.set("is_real", 1)

    def jal(self):
        k, p = self.rows_of("Jal")
        if len(p) == 0:
            return
        tb = self.table("Jal", len(p))
        self.fill_state(tb, k, p)
        self.fill_adapter(tb, k, p, "J")
        imm = self.ex.imm[p]
        tb.set("adapter.op_b_imm", limbs16(imm))
        pc = self.ex.pc(p)
        tb.set("next_pc", limbs16(pc + imm))
        rd0 = self.ex.slot_reg["A"][p] == 0
        tb.set("op_a_value", limbs16(pc + 4) * (~rd0).to(I64)[:, None])
        tb.set("is_real", 1)

    def jalr(self):
        k, p = self.rows_of("Jalr")
        if len(p) == 0:
            return
        tb = self.table("Jalr", len(p))
        self.fill_state(tb, k, p)
        _, bv, cv = self.fill_adapter(tb, k, p, "I")
        tb.set("next_pc", limbs16(bv + cv))
        tb.set("op_a_value", limbs16(self.ex.pc(p) + 4))
        tb.set("lsb", (bv + cv) & 1)
        tb.set("is_real", 1)

    def branch(self):
        k, p = self.rows_of("Branch")
        if len(p) == 0:
            return
        ex = self.ex
        tb = self.table("Branch", len(p))
        self.fill_state(tb, k, p)
        av, bv, _ = self.fill_adapter(tb, k, p, "I")
        op = ex.op[p]
        for nm in BRANCH_OPS:
            tb.set("is_" + nm.lower(), (op == OPC[nm]).to(I64))
        signed = (op == OPC["BLT"]) | (op == OPC["BGE"])
        self.fill_lt(tb, "cmp", av, bv, signed)
        eq = av == bv
        lt = torch.where(signed, av < bv, ult(av, bv))
        taken = torch.where(op == OPC["BEQ"], eq, torch.where(op == OPC["BNE"], ~eq,
                            torch.where((op == OPC["BLT"]) | (op == OPC["BLTU"]), lt, ~lt)))
        tb.set("is_branching", taken.to(I64))
        tb.set("next_pc", limbs16(ex.pc(p) + 4)[:, :3])          # taken or not, the offset is 4

    def memory_instructions(self):
        """Load / store chips' event_to_row (memory/instructions/**) + MemoryAccessCols::populate (consistency/trace.rs:L64-L101)."""
        ex, b, dev = self.ex, self.b, self.dev
        names = [c for c in list(LOAD_KINDS) + list(STORE_KINDS) if (b.chip == c).any()]
        self.mem_words = None
        if not names:
            return
        ks, ps = zip(*(self.rows_of(c) for c in names))
        k, p = torch.cat(ks), torch.cat(ps)
        op = ex.op[p]
        ptr = ex.reg_value(k, p, "B")
        addr = ptr + ex.imm[p]
        word = addr >> 3
        t_cur = ex.T(k, p) + POS_OFF["M"]
        is_store = torch.zeros_like(op, dtype=torch.bool)
        nbytes = torch.ones_like(op)
        for nm, nb in ACCESS_BYTES.items():
            m = op == OPC[nm]
            nbytes = torch.where(m, torch.full_like(op, nb), nbytes)
            if nm.startswith("S"):
                is_store |= m
        store_reg_val = ex.reg_value(k, p, "A")                           # stores: op_a = rs2
        # sort by (word, time)
        order = torch.argsort(t_cur, stable=True)
        order = order[torch.argsort(word[order], stable=True)]
        w_s, t_s = word[order], t_cur[order]
        first = torch.ones_like(w_s, dtype=torch.bool)
        first[1:] = w_s[1:] != w_s[:-1]
        prev_t = torch.where(first, torch.zeros_like(t_s), torch.roll(t_s, 1))
        # values: byte lanes, last-writer scan within each word's segment
        idx = torch.arange(len(w_s), device=dev)
        seg_start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), dim=0).values
        st_s, nb_s, off_s = is_store[order], nbytes[order], (addr[order] & 7)
        data_s = store_reg_val[order]
        initial = self.initial_word(w_s)
        after = torch.zeros_like(initial)
        for lane in range(8):
            covers = st_s & (off_s <= lane) & (lane < off_s + nb_s)
            lastw = torch.cummax(torch.where(covers, idx, torch.full_like(idx, -1)), dim=0).values
            has = lastw >= seg_start
            src = lastw.clamp(min=0)
            byte_from_store = (data_s[src] >> (8 * (lane - off_s[src]).clamp(min=0))) & 0xFF
            byte = torch.where(has, byte_from_store, (initial >> (8 * lane)) & 0xFF)
            after = after | (byte << (8 * lane))
        prev_val = torch.where(first, initial, torch.roll(after, 1))
        # un-sort
        inv = torch.empty_like(order)
        inv[order] = idx
        prev_t_u, prev_val_u, after_u = prev_t[inv], prev_val[inv], after[inv]
        # per word: initial value, final value, last timestamp (for MemoryLocal)
        last = torch.ones_like(first)
        last[:-1] = first[1:]
        self.mem_words = (w_s[last], initial[last], after[last], t_s[last])
        base = 0
        for c, kc in zip(names, ks):
            n = len(kc)
            sl = slice(base, base + n)
            base += n
            self.fill_mem_chip(c, k[sl], p[sl], addr[sl], t_cur[sl], prev_t_u[sl], prev_val_u[sl], after_u[sl], store_reg_val[sl])

    def initial_word(self, w):
        """Initial 64-bit content of word index w: the read-only image, or the store region's initial image (same generator)."""
        return self.ex.load_value(w << 3)

    def fill_mem_chip(self, chip, k, p, addr, t_cur, t_prev, prev_val, new_val, reg_val):
        ex = self.ex
        tb = self.table(chip, len(p))
        self.fill_state(tb, k, p)
        self.fill_adapter(tb, k, p, "I")
        al = limbs16(addr)
        tb.set("address.value", al[:, :3])
        tb.set("address.top_two_limb_inv", finv(al[:, 1] + al[:, 2]))
        tb.set("memory_access.prev_value", limbs16(prev_val))
        ph, pl, ch, cl = t_prev >> 24, t_prev & 0xFFFFFF, t_cur >> 24, t_cur & 0xFFFFFF
        same = ph == ch
        tb.set("memory_access.prev_high", ph)
        tb.set("memory_access.prev_low", pl)
        tb.set("memory_access.compare_low", same.to(I64))
        d = torch.where(same, cl - pl, ch - ph) - 1
        tb.set("memory_access.diff_low_limb", d & MASK16)
        tb.set("memory_access.diff_high_limb", d >> 16)
        op = ex.op[p]
        bits = [(addr >> i) & 1 for i in range(3)]
        pvl = limbs16(prev_val)
        rows = torch.arange(len(p), device=self.dev)
        if chip in ("LoadByte", "StoreByte", "LoadX0"):
            tb.set("offset_bit", torch.stack(bits, dim=1))
        elif chip in ("LoadHalf", "StoreHalf"):
            tb.set("offset_bit", torch.stack(bits[1:], dim=1))
        elif chip in ("LoadWord", "StoreWord"):
            tb.set("offset_bit", bits[2])
        limb = pvl[rows, (addr >> 1) & 3]
        if chip == "LoadByte":
            byte = (limb >> (8 * bits[0])) & 0xFF
            lb = op == OPC["LB"]
            tb.set("selected_limb", limb)
            tb.set("selected_limb_low_byte", limb & 0xFF)
            tb.set("selected_byte", byte)
            tb.set("msb", lb.to(I64) * (byte >> 7))
            tb.set("is_lb", lb.to(I64))
            tb.set("is_lbu", (~lb).to(I64))
        elif chip == "LoadHalf":
            lh = op == OPC["LH"]
            tb.set("selected_half", limb)
            tb.set("msb", lh.to(I64) * (limb >> 15))
            tb.set("is_lh", lh.to(I64))
            tb.set("is_lhu", (~lh).to(I64))
        elif chip == "LoadWord":
            lw = op == OPC["LW"]
            sel = torch.where(bits[2][:, None] == 1, pvl[:, 2:], pvl[:, :2])
            tb.set("selected_word", sel)
            tb.set("msb", lw.to(I64) * (sel[:, 1] >> 15))
            tb.set("is_lw", lw.to(I64))
            tb.set("is_lwu", (~lw).to(I64))
        elif chip == "LoadDouble":
            tb.set("is_real", 1)
        elif chip == "LoadX0":
            for nm in LOAD_KINDS["LoadX0"]:
                tb.set("is_" + nm.lower(), (op == OPC[nm]).to(I64))
        else:
            tb.set("is_real", 1)
            if chip != "StoreDouble":
                tb.set("store_value", limbs16(new_val))
            if chip == "StoreByte":
                rl = reg_val & 0xFF
                ml, mh = limb & 0xFF, limb >> 8
                tb.set("mem_limb", limb)
                tb.set("mem_limb_low_byte", ml)
                tb.set("register_low_byte", rl)
                inc = torch.where(bits[0] == 1, 256 * (rl - mh), rl - ml)
                tb.set("increment", inc % P)

    # -- the CPU state chain: StateBump rows where a sent state is not in normal form (adapter/bump.rs)
    def state_chain(self):
        ex, b, dev = self.ex, self.b, self.dev
        K, L = ex.K, ex.L
        n = torch.arange(K * L, device=dev)
        k, p = n // L, n % L
        T = ex.T(k, p)
        pc = ex.pc(p)
        normal_pc = torch.as_tensor(np.isin(b.chip, ["Branch", "Jal", "Jalr"]), device=dev)[p]   # these chips normalise next_pc
        nxt_pc = torch.where(p == L - 1, torch.full_like(pc, b.pc_base), pc + 4)
        pc_carry = (~normal_pc) & (((pc & MASK16) + 4) > MASK16)
        inc = ex.clk_inc[p]
        clk_carry = ((T & 0xFFFFFF) + inc) >= (1 << 24)
        need = pc_carry | clk_carry
        self.final_state = (int(T[-1]) + int(inc[-1]), int(nxt_pc[-1]))
        if bool(need.any()):
            Tn, pcn, nxt, pcc, incn = T[need], pc[need], nxt_pc[need], pc_carry[need], inc[need]
            air, _ = R.chip("StateBump")
            tb = Table(air, len(Tn), dev)
            self.tables["StateBump"] = tb
            nT = Tn + incn
            tb.set("next_clk_32_48", nT >> 32)
            tb.set("next_clk_24_32", (nT >> 24) & 0xFF)
            tb.set("next_clk_16_24", (nT >> 16) & 0xFF)
            tb.set("next_clk_0_16", nT & MASK16)
            tb.set("clk_high", Tn >> 24)
            tb.set("clk_low", (Tn & 0xFFFFFF) + incn)
            tb.set("next_pc", limbs16(nxt)[:, :3])
            sent = limbs16(pcn)[:, :3].clone()
            sent_norm = limbs16(nxt)[:, :3]
            sent[:, 0] += 4
            tb.set("pc", torch.where(pcc[:, None], sent, sent_norm))
            tb.set("is_clk", ((nT >> 24) != (Tn >> 24)).to(I64))
            tb.set("is_real", 1)

    def memory_local_and_bumps(self):
        """MemoryLocal rows (memory/local.rs generate_trace: one row per touched address) and MemoryBump rows (memory/bump.rs)."""
        ex, dev = self.ex, self.dev
        K = ex.K
        regs = torch.as_tensor(ex.touched_regs, device=dev)
        lp = torch.as_tensor([ex.last_access[int(r)][0] for r in regs], device=dev)
        lo = torch.as_tensor([ex.last_access[int(r)][1] for r in regs], device=dev)
        kk = torch.full_like(lp, K - 1)
        final_t = ex.T(kk, lp) + lo
        lastw = ex.lastw[regs]
        final_v = torch.where(lastw >= 0, ex.W[K - 1, lastw.clamp(min=0)], ex.init[regs])
        final_v = torch.where(regs == 0, torch.zeros_like(final_v), final_v)
        init_v = torch.where(regs == 0, torch.zeros_like(final_v), ex.init[regs])
        addr, iv, fv, ft = regs, init_v, final_v, final_t
        if self.mem_words is not None:
            w, wi, wf, wt = self.mem_words
            addr, iv, fv, ft = torch.cat([addr, w << 3]), torch.cat([iv, wi]), torch.cat([fv, wf]), torch.cat([ft, wt])
        air, _ = R.chip("MemoryLocal")
        tb = Table(air, len(addr), dev)
        self.tables["MemoryLocal"] = tb
        tb.set("addr", limbs16(addr)[:, :3])
        tb.set("final_clk_high", ft >> 24)
        tb.set("final_clk_low", ft & 0xFFFFFF)
        for tag, v in (("initial", iv), ("final", fv)):
            l = limbs16(v)
            tb.set(tag + "_value", l)
            tb.set(tag + "_value_lower", l[:, 2] & 0xFF)
            tb.set(tag + "_value_upper", l[:, 2] >> 8)
        tb.set("is_real", 1)
        self._bump_rows()

    def _bump_rows(self):
        if self.bumps:
            dev = self.dev
            reg, tp, val, tc = (torch.cat(x) for x in zip(*self.bumps))
            air, _ = R.chip("MemoryBump")
            tb = Table(air, len(reg), dev)
            self.tables["MemoryBump"] = tb
            tb.set("access.prev_value", limbs16(val))
            ph, pl, ch = tp >> 24, tp & 0xFFFFFF, tc >> 24
            tb.set("access.prev_high", ph)
            tb.set("access.prev_low", pl)
            tb.set("access.compare_low", 0)
            d = ch - ph - 1
            tb.set("access.diff_low_limb", d & MASK16)
            tb.set("access.diff_high_limb", d >> 16)
            tb.set("clk_32_48", tc >> 32)
            tb.set("clk_24_32", (tc >> 24) & 0xFF)
            tb.set("clk_16_24", 0)
            tb.set("clk_0_16", 0)
            tb.set("addr", reg)
            tb.set("is_real", 1)

    # -- public values, the cluster's empty chips, table multiplicities
    def finish(self):
        from . import public_values as PVM
        ex, b, dev = self.ex, self.b, self.dev
        machine = {}
        for name, tb in self.tables.items():
            machine[name] = R.chip(name)
        t0, t1 = ex.clk0, self.final_state[0]
        pc1 = self.final_state[1]
        if (t1 >> 24) != ((t1 - 8) >> 24) and "StateBump" not in self.tables:
            raise AssertionError("unreachable: a clock carry always has its StateBump row")
        # the shard's public values (record.rs postprocess + finalize_public_values): one execution shard from clk0 to the loop's end
        pv = PVM.no_memory_events(PVM.set_state(PVM.blank(), b.pc_base, pc1, t0, t1, 0, True))
        # the global interactions: MemoryLocal's two per row (initial = receive, final = send), then SyscallCore's
        ml = self.tables["MemoryLocal"]
        msgs = eval_interactions(R.chip("MemoryLocal")[1], ml.main[:ml.n], None, kinds=(R.GLOBAL,))
        (_, recv, _), (_, send, _) = msgs                              # MemoryLocal's two Global sends, [rows, 11] each
        events = [torch.stack([recv, send], dim=1).reshape(-1, 11)]    # per row: initial = receive, final = send
        for name in ("SyscallCore",):                                  # the other chips of a core shard that talk to the Global chip
            if name in self.tables:
                t_ = self.tables[name]
                events += [v for _, v, _ in eval_interactions(R.chip(name)[1], t_.main[:t_.n], None, kinds=(R.GLOBAL,))]
        if getattr(self, "real_global", True):
            PVM.set_global(pv, *self.global_chip(machine, torch.cat(events)))
        else:
            PVM.set_global(pv, 0, None)
            air, it = global_sink_chip()
            rows = torch.cat(events)
            tb = Table(air, rows.shape[0], dev)
            tb.main[:tb.n, :11] = rows
            tb.main[:tb.n, 11] = 1
            self.tables["GlobalSink"], machine["GlobalSink"] = tb, (air, it)
        # Program: one row per body position (program/trusted.rs:L80-L131), multiplicity = K
        air, it = R.chip("Program")
        tb = Table(air, ex.L, dev)
        p = torch.arange(ex.L, device=dev)
        self._program_rows(tb, p)
        tb.main[:tb.n, 0] = ex.K
        if tb.prep.shape[0] > tb.n:                               # padding rows repeat instruction 0 with multiplicity 0
            tb.prep[tb.n:] = tb.prep[0]
        self.tables["Program"], machine["Program"] = tb, (air, it)
        self.byte_range_tables(machine, pv)
        self.fill_cluster(machine, CORE_CLUSTER)
        names = sorted(machine)
        return [machine[n] for n in names], {n: (self.tables[n].prep, self.tables[n].main) for n in names}, PVM.to_tensor(pv)

    def fill_cluster(self, machine, cluster):
        """The chips of the shard's shape cluster that have no events: height zero (`PaddedMle::zeros`, hypercube/src/prover/
        trace.rs:L157-L180) — they are in the proof (opened values, LogUp-GKR interactions) like every other chip."""
        for name in cluster:
            if name not in machine:
                air, it = R.chip(name)
                assert air.prep_width == 0, name
                self.tables[name], machine[name] = EmptyTable(air, self.dev), (air, it)

    def byte_range_tables(self, machine, publics):
        """Byte / Range tables (bytes/trace.rs, range/trace.rs) with multiplicities COUNTED from the byte messages every table of
        `self.tables` sends on its rows and the ones `eval_public_values` sends for `publics` (the generate_dependencies of the
        two tables: bytes/trace.rs:L50-L66, range/trace.rs:L54-L95); every message is checked to be a row of the table it addresses."""
        from . import public_values as PVM
        dev = self.dev
        byte_air, byte_it = R.chip("Byte")
        range_air, range_it = R.chip("Range")
        bt, rt = Table(byte_air, 1 << 16, dev), Table(range_air, 1 << 17, dev)
        bc = torch.arange(1 << 16, device=dev)
        bb, cc = bc >> 8, bc & 0xFF
        bt.prep[:, 0], bt.prep[:, 1], bt.prep[:, 2], bt.prep[:, 3], bt.prep[:, 4] = bb, cc, bb & cc, bb | cc, bb ^ cc
        bt.prep[:, 5], bt.prep[:, 6] = (bb < cc).to(I64), bb >> 7
        ri = torch.arange(1 << 17, device=dev)
        bits = torch.where(ri == 0, torch.zeros_like(ri), (torch.log2(ri.clamp(min=1).to(torch.float64)).floor()).to(I64))
        rt.prep[:, 0], rt.prep[:, 1] = torch.where(ri == 0, torch.zeros_like(ri), ri - (1 << bits)), bits
        senders = [(name, machine[name][1], tbl.main[:tbl.n], tbl.prep[:tbl.n] if tbl.prep is not None else None) for name, tbl in self.tables.items()]
        senders.append(("PublicValues", PVM.program()[1], PVM.row(publics, dev), None))
        for name, it, main_rows, prep_rows in senders:
            for mult, vals, _ in eval_interactions(it, main_rows, prep_rows, kinds=(R.BYTE,), sends_only=True):
                opc, a, x, y = vals[:, 0], vals[:, 1], vals[:, 2], vals[:, 3]
                is_range = opc == R.B_RANGE
                if bool(is_range.any()):
                    aa, bits_ = a[is_range], x[is_range]
                    assert bool(((bits_ <= 16) & (aa < (1 << bits_.clamp(max=16))) & (y[is_range] == 0)).all()), (name, "range check fails")
                    tally(rt.main[:, 0], (1 << bits_) + aa, mult[is_range])
                nb = ~is_range
                if bool(nb.any()):
                    o, aa, xx, yy, mm = opc[nb], a[nb], x[nb], y[nb], mult[nb]
                    assert bool(((xx < 256) & (yy < 256) & (o < 6)).all()), (name, "byte operand out of range")
                    row = xx * 256 + yy
                    want = torch.where(o == R.B_U8RANGE, torch.zeros_like(aa),
                                       torch.where(o == R.B_MSB, xx >> 7, bt.prep[row, 2 + o.clamp(max=4) - (o > 3).to(I64)]))
                    assert bool((aa == want).all()), (name, "byte lookup result is wrong")
                    assert bool(((o != R.B_MSB) | (yy == 0)).all()), name
                    tally(bt.main.view(-1), row * 6 + o, mm)
        self.tables["Byte"], machine["Byte"] = bt, (byte_air, byte_it)
        self.tables["Range"], machine["Range"] = rt, (range_air, range_it)

    def global_chip(self, machine, ev):
        """GlobalChip::generate_trace_into (global/mod.rs:L131-L260): one row per global interaction event `ev` [n, 11] — the two
        events of every MemoryLocal row (memory/local.rs generate_dependencies: initial = receive, final = send), then the
        syscall events of SyscallCore. Returns (number of events, the digest they accumulate to [x[7], y[7]])."""
        from . import septic as SE
        dev = self.dev
        n = ev.shape[0]
        message, is_send, is_recv, kind = ev[:, :8], ev[:, 8], ev[:, 9], ev[:, 10]
        x, y, off, perm = SE.lift_x(message, kind, is_recv == 1)
        air, it = R.chip("Global")
        tb = Table(air, n, dev)
        tb.set("message", message)
        tb.set("kind", kind)
        tb.set("message_0_16bit_limb", message[:, 0] & MASK16)
        tb.set("message_0_8bit_limb", (message[:, 0] >> 16) & 0xFF)
        tb.set("interaction.x_coordinate", x)
        tb.set("interaction.y_coordinate", y)
        tb.set("interaction.offset", off)
        rc = torch.where(is_recv == 1, y[:, 6] - 1, P - y[:, 6] - 1)
        assert bool(((rc >= 0) & (rc < (63 << 24))).all())
        tb.set("interaction.y6_byte_decomp", torch.stack([rc & 0xFF, (rc >> 8) & 0xFF, (rc >> 16) & 0xFF, rc >> 24], dim=1))
        tb.set("is_real", 1)
        tb.set("is_receive", is_recv)
        tb.set("is_send", is_send)
        tb.set("index", torch.arange(n, device=dev))
        start = tuple(torch.tensor(v, dtype=I64, device=dev) for v in R.CURVE_CUMULATIVE_SUM_START)
        dummy = tuple(torch.tensor(v, dtype=I64, device=dev) for v in R.CURVE_WITNESS_DUMMY_POINT)
        cx, cy = SE.prefix_sums(start, x, y)
        tb.set("accumulation.initial_digest_x", torch.cat([start[0][None], cx[:-1]]))
        tb.set("accumulation.initial_digest_y", torch.cat([start[1][None], cy[:-1]]))
        tb.set("accumulation.cumulative_sum_x", cx)
        tb.set("accumulation.cumulative_sum_y", cy)
        pcol = tb.L["interaction.permutation"]
        tb.main[:n, pcol:pcol + perm.shape[1]] = perm
        if tb.main.shape[0] > n:                                       # padding rows: populate_dummy (global/mod.rs:L213-L236)
            pad = tb.main[n:]
            pad[:, tb.L["interaction.x_coordinate"]:tb.L["interaction.x_coordinate"] + 7] = dummy[0]
            pad[:, tb.L["interaction.y_coordinate"]:tb.L["interaction.y_coordinate"] + 7] = dummy[1]
            pad[:, pcol:pcol + perm.shape[1]] = SE.poseidon2_rows(torch.zeros((1, 16), dtype=I64, device=dev))[0]
            sdx, sdy = SE.ec_add((start[0][None], start[1][None]), (dummy[0][None], dummy[1][None]))
            for nm, v in (("initial_digest_x", start[0]), ("initial_digest_y", start[1]), ("cumulative_sum_x", sdx[0]), ("cumulative_sum_y", sdy[0])):
                c0 = tb.L["accumulation." + nm]
                pad[:, c0:c0 + 7] = v
        self.tables["Global"], machine["Global"] = tb, (air, it)
        # the two ends of the accumulation chain are public values: global_count, global_cumulative_sum (eval_global_sum)
        return n, [int(v) for v in cx[-1]] + [int(v) for v in cy[-1]]

    def _program_rows(self, tb, p):
        ex = self.ex
        pc = ex.pc(p)
        tb.prep[:tb.n, 0:3] = limbs16(pc)[:, :3]
        tb.prep[:tb.n, 3] = ex.op[p]
        A, Bq, C = ex.slot_reg["A"][p], ex.slot_reg["B"][p], ex.slot_reg["C"][p]
        op = ex.op[p]
        tb.prep[:tb.n, 4] = A.clamp(min=0)
        # op_b: a register number, or the immediate of the J / U types (op_b = op_c = imm for U; JAL: op_b = imm, op_c = 0)
        utype = (op == OPC["LUI"]) | (op == OPC["AUIPC"])
        jal = op == OPC["JAL"]
        imm_l = limbs16(ex.imm[p])
        regw = lambda r: torch.stack([r.clamp(min=0)] + [torch.zeros_like(r)] * 3, dim=1)
        tb.prep[:tb.n, 5:9] = torch.where((utype | jal)[:, None], imm_l, regw(Bq))
        opc = torch.where(ex.has_imm[p][:, None], imm_l, regw(C))
        opc = torch.where(jal[:, None], torch.zeros_like(opc), opc)
        tb.prep[:tb.n, 9:13] = opc
        tb.prep[:tb.n, 13] = (A == 0).to(I64)
        tb.prep[:tb.n, 14] = (utype | jal).to(I64)
        tb.prep[:tb.n, 15] = ex.has_imm[p].to(I64)


def tally(target, idx, mult):
    """target[idx] += mult without atomics (a few table rows — limb 0, byte pair (0, 0) — receive millions of messages, and
    float atomics on one address serialise on the GPU): sort, prefix-sum, difference at the run boundaries."""
    order = torch.argsort(idx)
    si, sm = idx[order], torch.cumsum(mult[order], dim=0)
    last = torch.ones_like(si, dtype=torch.bool)
    last[:-1] = si[1:] != si[:-1]
    ends, keys = sm[last], si[last]
    target[keys] += ends - torch.cat([ends.new_zeros(1), ends[:-1]])


def eval_vcol(v, prep, main):
    acc = torch.full((main.shape[0],), v.constant, dtype=I64, device=main.device)
    for kind, col, w in v.terms:
        acc = (acc + (main if kind == "main" else prep)[:, col] * w) % P
    return acc


def eval_interactions(it, main, prep, kinds=None, sends_only=False):
    """[(multiplicity [m], values [m, n_values], is_send)] over the rows with a non-zero multiplicity."""
    out = []
    for is_send, lst in ((True, it.sends), (False, it.receives)):
        if sends_only and not is_send:
            continue
        for kind, values, mult in lst:
            if kinds is not None and kind not in kinds:
                continue
            m = eval_vcol(mult, prep, main)
            live = m != 0
            if not bool(live.any()):
                continue
            mm, pl = main[live], (prep[live] if prep is not None else None)
            vals = torch.stack([eval_vcol(v, pl, mm) for v in values], dim=1)
            out.append((m[live], vals, is_send))
    return out


def global_sink_chip():
    """Synthetic: receives MemoryLocal's 11-word `Global` messages where the reference's Global chip would."""
    from .rv_builder import Builder
    b = Builder("GlobalSink", 12)
    c = [b.main(i) for i in range(12)]
    b.assert_bool(c[11])
    b.receive(R.GLOBAL, c[:11], c[11])
    return b.air, b.it


def to_monty_np(t):
    """canonical int64 tensor -> Montgomery uint32 numpy (row-major)."""
    return ((t.cpu().numpy().astype(np.uint64) << np.uint64(32)) % np.uint64(P)).astype(np.uint32)
