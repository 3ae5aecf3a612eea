"""RISC-V (rv64im) core chips of SP1 v6 as DATA: constraints + interactions, hand-transcribed.

The reference's chips are Rust `Air::eval` bodies (/root/reference/crates/core/machine/src/**); no Rust toolchain exists in
this image, so — exactly as `recursion.py` did for the recursion machine — each `eval` (Supervisor mode: `M::IS_TRUSTED`,
`mprotect` feature off, which is what `rv64im_complexity.json` counts) is transcribed against the recording builder of
`rv_builder.py`, operation by operation, in the reference's call order:

    chip            width  reference eval
    Add               33   alu/add_sub/add.rs:L203-L283        Addi     30  alu/add_sub/addi.rs:L202-L290
    Sub               33   alu/add_sub/sub.rs:L202-L288        Bitwise  51  alu/bitwise/mod.rs:L209-L358
    Lt                44   alu/lt/mod.rs:L213-L348             Mul      82  alu/mul/mod.rs:L229-L385
    ShiftLeft         65   alu/sll/mod.rs:L304-L543            ShiftRight 69 alu/sr/mod.rs:L330-L665
    UType             31   utype/mod.rs:L85-L202               ...      (see CHIPS at the end for the full list)
    MemoryLocal       20   memory/local.rs:L257-L360           MemoryBump 15 memory/bump.rs:L186-L209
    StateBump         14   adapter/bump.rs:L185-L248           Program 1+16 program/trusted.rs:L311-L323
    Byte             6+7   bytes/air.rs:L23-L55                Range   1+2  range/air.rs:L21-L34

  shared operations: adapter/state.rs (CPUState), adapter/register/{r,i,j,alu}_type.rs, air/memory.rs (register / memory
  access arguments), operations/{add,sub,bitwise,bitwise_u16,u16_operation,slt,u16_compare,msb,mul,...}.rs.

What pins the transcription (there is no reference-made RISC-V proof in the tree, unlike the recursion machine):
  * column counts == /root/reference/crates/core/executor/src/artifacts/rv64im_costs.json,
  * `assert_zero` counts == rv64im_complexity.json (`chip.num_constraints`, riscv/mod.rs:L1843-L1863),
  * interaction counts == the recorded core shard 0 of /root/reference/sp1-gpu/crates/logup_gkr/layer_workloads.json
    (chips in name order; its row counts are rows / 4: Byte 16384 = 65536 / 4, Range 32768 = 2^17 / 4) —
    three independent data points per chip, checked by tests/test_riscv_machine.py from the committed table RECORDED below,
  * and semantics: `riscv_trace.py` EXECUTES random rv64im code and fills the columns from the executed values; every
    constraint must vanish on every row and every bus (Byte, Memory, Program, State, Global) must balance as a multiset
    (tests/machine_check.py) — a wrong sign, limb or carry in a transcription fails there.
"""
from types import SimpleNamespace

from ..air import P
from .rv_builder import Builder, Cols

# InteractionKind (hypercube/src/lookup/interaction.rs:L27-L80)
MEMORY, PROGRAM, BYTE, STATE, SYSCALL, GLOBAL, GLOBAL_ACC = 1, 2, 5, 7, 8, 9, 13
# ByteOpcode (core/executor/src/opcode.rs:L163-L178)
B_AND, B_OR, B_XOR, B_U8RANGE, B_LTU, B_MSB, B_RANGE = range(7)
# Opcode (core/executor/src/opcode.rs:L46-L153)
OPC = {n: i for i, n in enumerate(
    "ADD ADDI SUB XOR OR AND SLL SRL SRA SLT SLTU MUL MULH MULHU MULHSU DIV DIVU REM REMU ADDW SUBW SLLW SRLW SRAW MULW DIVW "
    "DIVUW REMW REMUW LB LH LW LBU LHU LWU LD SB SH SW SD BEQ BNE BLT BGE BLTU BGEU JAL JALR AUIPC LUI ECALL EBREAK UNIMP".split())}
PC_INC, CLK_INC = 4, 8
POS_MEMORY, POS_C, POS_B, POS_A = 1, 2, 3, 4          # MemoryAccessPosition (core/executor/src/events/memory.rs:L63-L74)
INV = lambda k: pow(k, -1, P)

# (columns, constraints, interactions) per chip: rv64im_costs.json / rv64im_complexity.json / layer_workloads.json shard 0
RECORDED = {
    "Add": (33, 16, 21), "Addi": (30, 15, 17), "Addw": (36, 21, 20), "Bitwise": (51, 20, 25), "Branch": (45, 49, 19),
    "Byte": (13, 0, 6), "DivRem": (246, 348, 135), "Global": (241, 216, 7), "Jal": (31, 24, 17), "Jalr": (35, 26, 21),
    "LoadByte": (47, 32, 23), "LoadDouble": (39, 24, 21), "LoadHalf": (44, 33, 22), "LoadWord": (44, 33, 22),
    "LoadX0": (48, 35, 21), "Lt": (44, 42, 20), "MemoryBump": (15, 5, 7), "MemoryLocal": (20, 4, 14), "Mul": (82, 61, 52),
    "Program": (17, 0, 1), "Range": (3, 0, 1), "ShiftLeft": (65, 69, 27), "ShiftRight": (69, 78, 29), "StateBump": (14, 8, 8),
    "StoreByte": (50, 32, 23), "StoreDouble": (39, 23, 21), "StoreHalf": (45, 27, 21), "StoreWord": (44, 27, 21),
    "Sub": (33, 16, 21), "Subw": (32, 16, 20), "SyscallCore": (10, 2, 4), "SyscallInstrs": (65, 93, 30), "UType": (31, 19, 13),
}
# rows of that recorded shard (4 x the recorded count), in chip name order; Bitwise / Branch identified by their interaction counts
RECORDED_ROWS = {
    "Add": 56864, "Addi": 1451424, "Addw": 1248, "Bitwise": 517920, "Branch": 743392, "Byte": 65536, "DivRem": 8,
    "Global": 494560, "Jal": 13440, "Jalr": 141600, "LoadByte": 649088, "LoadDouble": 688544, "LoadHalf": 800,
    "LoadWord": 9888, "LoadX0": 9536, "Lt": 18272, "MemoryBump": 64, "MemoryLocal": 247296, "Mul": 128, "Program": 472032,
    "Range": 131072, "ShiftLeft": 177216, "ShiftRight": 27712, "StateBump": 32, "StoreByte": 468896, "StoreDouble": 724544,
    "StoreHalf": 12128, "StoreWord": 10560, "Sub": 18592, "Subw": 32, "SyscallCore": 8, "SyscallInstrs": 32, "UType": 116640,
}


# ---------------------------------------------------------------------------------------------------------------------
# column structs (`#[repr(C)]`, field order = column order)
def S(*fields):
    """A struct layout: fields are (name, count | nested layout); returns alloc(cols, prefix) -> namespace of Syms."""
    def alloc(c, prefix=""):
        out = SimpleNamespace()
        for name, what in fields:
            full = prefix + name
            if callable(what):
                v = what(c, full + ".")
            elif what == 1:
                v = c.one(full)
            else:
                v = c.arr(what, full)
            setattr(out, name, v)
        return out
    return alloc


CPU_STATE = S(("clk_high", 1), ("clk_16_24", 1), ("clk_0_16", 1), ("pc", 3))                        # adapter/state.rs:L26-L32
REG_ACCESS = S(("prev_value", 4), ("prev_low", 1), ("diff_low_limb", 1))                             # memory/consistency/columns.rs:L60-L78
MEM_ACCESS = S(("prev_value", 4), ("prev_high", 1), ("prev_low", 1), ("compare_low", 1), ("diff_low_limb", 1),
               ("diff_high_limb", 1))                                                                 # columns.rs:L10-L35
R_TYPE = S(("op_a", 1), ("op_a_memory", REG_ACCESS), ("op_a_0", 1), ("op_b", 1), ("op_b_memory", REG_ACCESS), ("op_c", 1),
           ("op_c_memory", REG_ACCESS))                                                               # r_type.rs:L33-L41
I_TYPE = S(("op_a", 1), ("op_a_memory", REG_ACCESS), ("op_a_0", 1), ("op_b", 1), ("op_b_memory", REG_ACCESS), ("op_c_imm", 4))
J_TYPE = S(("op_a", 1), ("op_a_memory", REG_ACCESS), ("op_a_0", 1), ("op_b_imm", 4), ("op_c_imm", 4))
ALU_TYPE = S(("op_a", 1), ("op_a_memory", REG_ACCESS), ("op_a_0", 1), ("op_b", 1), ("op_b_memory", REG_ACCESS), ("op_c", 4),
             ("op_c_memory", REG_ACCESS), ("imm_c", 1))                                               # alu_type.rs:L33-L42
U16_TO_U8 = S(("low_bytes", 4),)
LT_UNSIGNED = S(("bit", 1), ("u16_flags", 4), ("not_eq_inv", 1), ("comparison_limbs", 2))            # slt.rs:L29-L34 (bit = u16_compare_operation)
LT_SIGNED = S(("result", LT_UNSIGNED), ("b_msb", 1), ("c_msb", 1))
ADDRESS_OP = S(("value", 3), ("top_two_limb_inv", 1))                                              # operations/address.rs:L28-L32
MUL_OP = S(("carry", 16), ("product", 16), ("b_lower_byte", U16_TO_U8), ("c_lower_byte", U16_TO_U8), ("b_msb", 1), ("c_msb", 1),
           ("product_msb", 1), ("b_sign_extend", 1), ("c_sign_extend", 1))                            # operations/mul.rs:L33-L51


def _chip(name, main_width, prep_width=0):
    b = Builder(name, main_width, prep_width)
    return b, Cols(b), (Cols(b, prep=True) if prep_width else None)


def _done(b, c, cp=None):
    assert c.n == b.air.main_width, (b.name, c.n, b.air.main_width)
    if cp is not None:
        assert cp.n == b.air.prep_width, (b.name, cp.n)
    b.air.layout, b.air.prep_layout = dict(c.names), (dict(cp.names) if cp is not None else {})
    return b.air, b.it


# ---------------------------------------------------------------------------------------------------------------------
# builder helpers of the reference
def send_byte(b, opcode, a, x, y, mult):                                   # hypercube/src/air/builder.rs:L113-L129
    b.send(BYTE, [opcode, a, x, y], mult)


def slice_range_check_u8(b, xs, mult):                                     # air/word.rs:L55-L78
    xs = list(xs)
    i = 0
    while i + 1 < len(xs):
        send_byte(b, B_U8RANGE, 0, xs[i], xs[i + 1], mult)
        i += 2
    if i < len(xs):
        send_byte(b, B_U8RANGE, 0, xs[i], 0, mult)


def slice_range_check_u16(b, xs, mult):                                    # air/word.rs:L80-L96
    for x in xs:
        send_byte(b, B_RANGE, x, 16, 0, mult)


def send_state(b, clk_high, clk_low, pc, mult):
    b.send(STATE, [clk_high, clk_low] + list(pc), mult)


def receive_state(b, clk_high, clk_low, pc, mult):
    b.receive(STATE, [clk_high, clk_low] + list(pc), mult)


def instruction_values(opcode, op_a, op_b, op_c, op_a_0, imm_b, imm_c):    # program/instruction.rs:L49-L63
    return [opcode, op_a] + list(op_b) + list(op_c) + [op_a_0, imm_b, imm_c]


def send_program(b, pc, instr, mult):                                      # air/program.rs:L12-L28
    b.send(PROGRAM, list(pc) + instr, mult)


def eval_register_access_timestamp(b, acc, do_check, clk):                 # air/memory.rs:L305-L334
    diff_minus_one = clk - acc.prev_low - 1
    diff_high_limb = (diff_minus_one - acc.diff_low_limb) * INV(1 << 16)
    send_byte(b, B_RANGE, acc.diff_low_limb, 16, 0, do_check)
    send_byte(b, B_U8RANGE, 0, diff_high_limb, 0, do_check)


def eval_register_access(b, clk_high, clk_low, addr, acc, write_value, do_check):
    """eval_register_access_write (air/memory.rs:L176-L222); a read is a write of prev_value (L124-L172)."""
    b.assert_bool(do_check)
    eval_register_access_timestamp(b, acc, do_check, clk_low)
    b.send(MEMORY, [clk_high, acc.prev_low] + list(addr) + list(acc.prev_value), do_check)
    b.receive(MEMORY, [clk_high, clk_low] + list(addr) + list(write_value), do_check)


def eval_memory_access_timestamp(b, acc, do_check, clk_high, clk_low):     # air/memory.rs:L255-L303
    b.when(do_check).assert_bool(acc.compare_low)
    b.when(do_check).when(acc.compare_low).assert_eq(clk_high, acc.prev_high)
    prev_comp = b.if_else(acc.compare_low, acc.prev_low, acc.prev_high)
    cur_comp = b.if_else(acc.compare_low, clk_low, clk_high)
    diff_minus_one = cur_comp - prev_comp - 1
    b.when(do_check).assert_eq(diff_minus_one, acc.diff_low_limb + acc.diff_high_limb * (1 << 16))
    send_byte(b, B_RANGE, acc.diff_low_limb, 16, 0, do_check)
    send_byte(b, B_U8RANGE, 0, acc.diff_high_limb, 0, do_check)


def eval_memory_access(b, clk_high, clk_low, addr, acc, write_value, do_check):
    """eval_memory_access_write (air/memory.rs:L69-L120); a read writes prev_value back (L19-L65)."""
    b.assert_bool(do_check)
    eval_memory_access_timestamp(b, acc, do_check, clk_high, clk_low)
    b.send(MEMORY, [acc.prev_high, acc.prev_low] + list(addr) + list(acc.prev_value), do_check)
    b.receive(MEMORY, [clk_high, clk_low] + list(addr) + list(write_value), do_check)


def clk_low_of(st):
    return st.clk_0_16 + st.clk_16_24 * (1 << 16)


def eval_cpu_state(b, st, next_pc, clk_increment, is_real):               # adapter/state.rs:L71-L100
    clk_high, clk_low = st.clk_high, clk_low_of(st)
    b.assert_bool(is_real)
    receive_state(b, clk_high, clk_low, st.pc, is_real)
    send_state(b, clk_high, clk_low + clk_increment, next_pc, is_real)
    send_byte(b, B_RANGE, (st.clk_0_16 - 1) * INV(8), 13, 0, is_real)
    slice_range_check_u8(b, [st.clk_16_24, b.const(0)], is_real)


def next_pc_inc(st):
    return [st.pc[0] + PC_INC, st.pc[1], st.pc[2]]


def _reg_addr(b, r):
    return [r, b.const(0), b.const(0)]


def _word_of(b, x):
    return [x, b.const(0), b.const(0), b.const(0)]


def eval_r_type(b, st, opcode, a_write, ad, is_real, is_trusted):          # r_type.rs:L86-L130
    clk_high, clk_low = st.clk_high, clk_low_of(st)
    b.assert_bool(is_real)
    send_program(b, st.pc, instruction_values(opcode, ad.op_a, _word_of(b, ad.op_b), _word_of(b, ad.op_c), ad.op_a_0, 0, 0), is_trusted)
    b.when(ad.op_a_0).assert_word_eq(a_write, [0, 0, 0, 0])
    eval_register_access(b, clk_high, clk_low + POS_A, _reg_addr(b, ad.op_a), ad.op_a_memory, a_write, is_real)
    eval_register_access(b, clk_high, clk_low + POS_B, _reg_addr(b, ad.op_b), ad.op_b_memory, ad.op_b_memory.prev_value, is_real)
    eval_register_access(b, clk_high, clk_low + POS_C, _reg_addr(b, ad.op_c), ad.op_c_memory, ad.op_c_memory.prev_value, is_real)


def eval_i_type(b, st, opcode, a_write, ad, is_real, is_trusted):          # i_type.rs:L82-L115
    clk_high, clk_low = st.clk_high, clk_low_of(st)
    b.assert_bool(is_real)
    send_program(b, st.pc, instruction_values(opcode, ad.op_a, _word_of(b, ad.op_b), ad.op_c_imm, ad.op_a_0, 0, 1), is_trusted)
    b.when(ad.op_a_0).assert_word_eq(a_write, [0, 0, 0, 0])
    eval_register_access(b, clk_high, clk_low + POS_A, _reg_addr(b, ad.op_a), ad.op_a_memory, a_write, is_real)
    eval_register_access(b, clk_high, clk_low + POS_B, _reg_addr(b, ad.op_b), ad.op_b_memory, ad.op_b_memory.prev_value, is_real)


def eval_j_type(b, st, opcode, a_write, ad, is_real, is_trusted):          # j_type.rs:L73-L98
    clk_high, clk_low = st.clk_high, clk_low_of(st)
    b.assert_bool(is_real)
    send_program(b, st.pc, instruction_values(opcode, ad.op_a, ad.op_b_imm, ad.op_c_imm, ad.op_a_0, 1, 1), is_trusted)
    b.when(ad.op_a_0).assert_word_eq(a_write, [0, 0, 0, 0])
    eval_register_access(b, clk_high, clk_low + POS_A, _reg_addr(b, ad.op_a), ad.op_a_memory, a_write, is_real)


def eval_alu_type(b, st, opcode, a_write, ad, is_real, is_trusted):        # alu_type.rs:L99-L142
    clk_high, clk_low = st.clk_high, clk_low_of(st)
    b.assert_bool(is_real)
    b.when_not(is_real).assert_eq(ad.imm_c, 0)
    send_program(b, st.pc, instruction_values(opcode, ad.op_a, _word_of(b, ad.op_b), ad.op_c, ad.op_a_0, 0, ad.imm_c), is_trusted)
    b.when(ad.op_a_0).assert_word_eq(a_write, [0, 0, 0, 0])
    eval_register_access(b, clk_high, clk_low + POS_A, _reg_addr(b, ad.op_a), ad.op_a_memory, a_write, is_real)
    eval_register_access(b, clk_high, clk_low + POS_B, _reg_addr(b, ad.op_b), ad.op_b_memory, ad.op_b_memory.prev_value, is_real)
    eval_register_access(b, clk_high, clk_low + POS_C, _reg_addr(b, ad.op_c[0]), ad.op_c_memory, ad.op_c_memory.prev_value,
                         is_real - ad.imm_c)
    b.when(ad.imm_c).assert_word_eq(ad.op_c_memory.prev_value, ad.op_c)


# ---------------------------------------------------------------------------------------------------------------------
# operations
def eval_add(b, x, y, value, is_real):                                     # operations/add.rs:L47-L73
    b.assert_bool(is_real)
    w, carry = b.when(is_real), b.const(0)
    for i in range(4):
        carry = (x[i] + y[i] - value[i] + carry) * INV(1 << 16)
        w.assert_bool(carry)
    slice_range_check_u16(b, value, is_real)


def eval_sub(b, x, y, value, is_real):                                     # operations/sub.rs:L45-L68
    b.assert_bool(is_real)
    w, carry = b.when(is_real), b.const(1)
    for i in range(4):
        carry = (x[i] + (1 << 16) - 1 - y[i] - value[i] + carry) * INV(1 << 16)
        w.assert_bool(carry)
    slice_range_check_u16(b, value, is_real)


def u16_to_u8_unsafe(b, limbs, low_bytes):                                 # operations/u16_operation.rs:L42-L56
    out = []
    for i in range(4):
        out += [low_bytes[i], (limbs[i] - low_bytes[i]) * INV(1 << 8)]
    return out


def u16_to_u8_safe(b, limbs, low_bytes, is_real):                          # u16_operation.rs:L58-L67
    out = u16_to_u8_unsafe(b, limbs, low_bytes)
    slice_range_check_u8(b, out, is_real)
    return out


def eval_msb(b, a, msb, is_real):                                          # operations/msb.rs:L47-L64
    b.assert_bool(is_real)
    b.assert_bool(msb)
    send_byte(b, B_RANGE, 2 * a - msb * (1 << 16), 16, 0, is_real)


def eval_compare_u16(b, x, y, bit, is_real):                               # operations/u16_compare.rs:L43-L60
    b.assert_bool(is_real)
    b.assert_bool(bit)
    send_byte(b, B_RANGE, x - y + bit * (1 << 16), 16, 0, is_real)


def eval_lt_unsigned(b, x, y, c, is_real):                                 # operations/slt.rs:L176-L231
    b.assert_bool(is_real)
    f = c.u16_flags
    sum_flags = f[0] + f[1] + f[2] + f[3]
    for i in range(4):
        b.assert_bool(f[i])
    b.assert_bool(sum_flags)
    is_comp_eq = 1 - sum_flags
    visited, bl, cl = b.const(0), b.const(0), b.const(0)
    for i in (3, 2, 1, 0):
        visited = visited + f[i]
        b.when(is_real - visited).assert_eq(x[i], y[i])
        bl = bl + x[i] * f[i]
        cl = cl + y[i] * f[i]
    b.assert_eq(bl, c.comparison_limbs[0])
    b.assert_eq(cl, c.comparison_limbs[1])
    b.when_not(is_comp_eq).assert_eq(c.not_eq_inv * (c.comparison_limbs[0] - c.comparison_limbs[1]), is_real)
    eval_compare_u16(b, c.comparison_limbs[0], c.comparison_limbs[1], c.bit, is_real)


def eval_lt_signed(b, x, y, c, is_signed, is_real):                        # operations/slt.rs:L82-L133
    b.assert_bool(is_signed)
    b.assert_bool(is_real)
    b.when_not(is_real).assert_zero(is_signed)
    eval_msb(b, x[3], c.b_msb, is_signed)
    eval_msb(b, y[3], c.c_msb, is_signed)
    b.when_not(is_signed).assert_zero(c.b_msb)
    b.when_not(is_signed).assert_zero(c.c_msb)
    xc, yc = list(x), list(y)
    xc[3] = x[3] + is_signed * (1 << 15) - (1 << 16) * c.b_msb
    yc[3] = y[3] + is_signed * (1 << 15) - (1 << 16) * c.c_msb
    eval_lt_unsigned(b, xc, yc, c.result, is_real)


def eval_mul(b, a, x, y, c, is_real, is_mul, is_mulh, is_mulw, is_mulhu, is_mulhsu, hint_products=None):   # operations/mul.rs:L140-L302
    xb = u16_to_u8_safe(b, x, c.b_lower_byte.low_bytes, is_real)
    yb = u16_to_u8_safe(b, y, c.c_lower_byte.low_bytes, is_real)
    for msb, byte in ((c.b_msb, xb[7]), (c.c_msb, yb[7])):
        send_byte(b, B_MSB, msb, byte, 0, is_real)
    eval_msb(b, a[1], c.product_msb, is_mulw)
    b.assert_eq(c.b_sign_extend, (is_mulh + is_mulhsu) * c.b_msb)
    b.assert_eq(c.c_sign_extend, is_mulh * c.c_msb)
    xe = xb + [c.b_sign_extend * 0xFF] * 8
    ye = yb + [c.c_sign_extend * 0xFF] * 8
    m = [b.const(0)] * 16
    for i in range(16):
        for j in range(16):
            if i + j < 16:
                m[i + j] = m[i + j] + xe[i] * ye[j]
    if hint_products is not None:          # the chip tells a prover that the next 16 asserts are these products (air.hint_mul)
        hint_products()
    for i in range(16):
        if i == 0:
            b.when(is_real).assert_eq(c.product[i], m[i] - c.carry[i] * (1 << 8))
        else:
            b.when(is_real).assert_eq(c.product[i], m[i] + c.carry[i - 1] - c.carry[i] * (1 << 8))
    is_upper = is_mulh + is_mulhu + is_mulhsu
    for i in range(4):
        if i < 2:
            b.when(is_mulw).assert_eq(c.product[2 * i] + c.product[2 * i + 1] * (1 << 8), a[i])
        else:
            b.when(is_mulw).assert_eq(c.product_msb * 0xFFFF, a[i])
        b.when(is_mul).assert_eq(c.product[2 * i] + c.product[2 * i + 1] * (1 << 8), a[i])
        b.when(is_upper).assert_eq(c.product[2 * i + 8] + c.product[2 * i + 9] * (1 << 8), a[i])
    for x_ in (c.b_msb, c.c_msb, c.b_sign_extend, c.c_sign_extend, is_mul, is_mulh, is_mulhu, is_mulhsu, is_mulw,
               is_mul + is_mulh + is_mulhu + is_mulhsu + is_mulw, is_real):
        b.assert_bool(x_)
    b.when(c.b_sign_extend).assert_eq(c.b_msb, 1)
    b.when(c.c_sign_extend).assert_eq(c.c_msb, 1)
    slice_range_check_u16(b, c.carry, is_real)
    slice_range_check_u8(b, c.product, is_real)


def eval_addw(b, x, y, value, msb, is_real):                               # operations/addw.rs:L33-L59
    b.assert_bool(is_real)
    w, carry = b.when(is_real), b.const(0)
    for i in range(2):
        carry = (x[i] + y[i] - value[i] + carry) * INV(1 << 16)
        w.assert_bool(carry)
    slice_range_check_u16(b, value, is_real)
    eval_msb(b, value[1], msb, is_real)


def eval_subw(b, x, y, value, msb, is_real):                               # operations/subw.rs:L37-L64
    b.assert_bool(is_real)
    w, carry = b.when(is_real), b.const(1)
    for i in range(2):
        carry = (x[i] + (1 << 16) - 1 - y[i] - value[i] + carry) * INV(1 << 16)
        w.assert_bool(carry)
    slice_range_check_u16(b, value, is_real)
    eval_msb(b, value[1], msb, is_real)


def eval_addr_add(b, x, y, value, is_real):                                # operations/addrs_add.rs:L41-L63
    b.assert_bool(is_real)
    w, carry = b.when(is_real), b.const(0)
    for i in range(4):
        v = value[i] if i < 3 else b.const(0)
        carry = (x[i] + y[i] - v + carry) * INV(1 << 16)
        w.assert_bool(carry)
    slice_range_check_u16(b, value, is_real)


def eval_address(b, x, y, bit0, bit1, bit2, is_real, c):                   # operations/address.rs:L47-L98
    bit0, bit1, bit2 = b._s(bit0), b._s(bit1), b._s(bit2)
    b.assert_bool(is_real)
    b.assert_bool(bit0)
    b.assert_bool(bit1)
    b.assert_bool(bit2)
    eval_addr_add(b, x, y, c.value, is_real)
    addr = c.value
    b.assert_eq(c.top_two_limb_inv * (addr[1] + addr[2]), is_real)
    aligned0 = addr[0] - 4 * bit2 - 2 * bit1 - bit0
    send_byte(b, B_RANGE, aligned0 * INV(8), 13, 0, is_real)
    return [aligned0, addr[1], addr[2]]


# ---------------------------------------------------------------------------------------------------------------------
# chips
def add_chip():
    b, c, _ = _chip("Add", 33)
    L = S(("state", CPU_STATE), ("adapter", R_TYPE), ("value", 4), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_add(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_memory.prev_value, L.value, L.is_real)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_r_type(b, L.state, OPC["ADD"], L.value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def addi_chip():
    b, c, _ = _chip("Addi", 30)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("value", 4), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_add(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, L.value, L.is_real)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_i_type(b, L.state, OPC["ADDI"], L.value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def sub_chip():
    b, c, _ = _chip("Sub", 33)
    L = S(("state", CPU_STATE), ("adapter", R_TYPE), ("value", 4), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_sub(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_memory.prev_value, L.value, L.is_real)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_r_type(b, L.state, OPC["SUB"], L.value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def bitwise_chip():
    b, c, _ = _chip("Bitwise", 51)
    L = S(("state", CPU_STATE), ("adapter", ALU_TYPE), ("b_low_bytes", U16_TO_U8), ("c_low_bytes", U16_TO_U8), ("result", 8),
          ("is_xor", 1), ("is_or", 1), ("is_and", 1))(c)
    is_real = L.is_xor + L.is_or + L.is_and
    for x in (L.is_xor, L.is_or, L.is_and, is_real):
        b.assert_bool(x)
    byte_opcode = L.is_xor * B_XOR + L.is_or * B_OR + L.is_and * B_AND
    cpu_opcode = L.is_xor * OPC["XOR"] + L.is_or * OPC["OR"] + L.is_and * OPC["AND"]
    b.assert_zero(L.adapter.op_a_0)
    # BitwiseU16Operation (operations/bitwise_u16.rs:L50-L92) over BitwiseOperation (operations/bitwise.rs:L53-L70)
    b.assert_bool(is_real)
    xb = u16_to_u8_unsafe(b, L.adapter.op_b_memory.prev_value, L.b_low_bytes.low_bytes)
    yb = u16_to_u8_unsafe(b, L.adapter.op_c_memory.prev_value, L.c_low_bytes.low_bytes)
    for i in range(8):
        send_byte(b, byte_opcode, L.result[i], xb[i], yb[i], is_real)
    result = [L.result[2 * i] + L.result[2 * i + 1] * (1 << 8) for i in range(4)]
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    eval_alu_type(b, L.state, cpu_opcode, result, L.adapter, is_real, is_real)
    return _done(b, c)


def lt_chip():
    b, c, _ = _chip("Lt", 44)
    L = S(("state", CPU_STATE), ("adapter", ALU_TYPE), ("is_slt", 1), ("is_sltu", 1), ("lt", LT_SIGNED))(c)
    is_real = L.is_slt + L.is_sltu
    for x in (L.is_slt, L.is_sltu, is_real):
        b.assert_bool(x)
    b.assert_zero(L.adapter.op_a_0)
    eval_lt_signed(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_memory.prev_value, L.lt, L.is_slt, is_real)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    opcode = L.is_slt * OPC["SLT"] + L.is_sltu * OPC["SLTU"]
    eval_alu_type(b, L.state, opcode, _word_of(b, L.lt.result.bit), L.adapter, is_real, is_real)
    return _done(b, c)


def mul_chip():
    b, c, _ = _chip("Mul", 82)
    L = S(("state", CPU_STATE), ("adapter", R_TYPE), ("a", 4), ("mul", MUL_OP), ("is_mul", 1), ("is_mulh", 1), ("is_mulhu", 1),
          ("is_mulhsu", 1), ("is_mulw", 1))(c)
    is_real = L.is_mul + L.is_mulh + L.is_mulhu + L.is_mulhsu + L.is_mulw
    # MulCols is #[repr(C)]: the MulOperation struct, then the five flags whose sum is is_real (alu/mul/mod.rs:L41-L68)
    hint = lambda: b.air.hint_mul(c.names["mul.carry"], c.names["adapter.op_b_memory.prev_value"])
    eval_mul(b, L.a, L.adapter.op_b_memory.prev_value, L.adapter.op_c_memory.prev_value, L.mul, is_real, L.is_mul, L.is_mulh,
             L.is_mulw, L.is_mulhu, L.is_mulhsu, hint_products=hint)
    for x in (L.is_mul, L.is_mulh, L.is_mulhu, L.is_mulw, L.is_mulhsu, is_real):
        b.assert_bool(x)
    opcode = (L.is_mul * OPC["MUL"] + L.is_mulh * OPC["MULH"] + L.is_mulhu * OPC["MULHU"] + L.is_mulhsu * OPC["MULHSU"]
              + L.is_mulw * OPC["MULW"])
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_r_type(b, L.state, opcode, L.a, L.adapter, is_real, is_real)
    return _done(b, c)


def _shift_common(b, L, is_real, not_word_sel, left):
    """The part ShiftLeft (alu/sll/mod.rs:L319-L396) and ShiftRight (alu/sr/mod.rs:L431-L527) share: the shift amount's
    bits, the limb selector, v = 2^(c mod 16) (left) or 2^(16 - c mod 16) (right), the split of every limb of b."""
    for i in range(6):
        b.assert_bool(L.c_bits[i])
    c_lower_bits, bit_shift = b.const(0), None
    for i in range(6):
        c_lower_bits = c_lower_bits + L.c_bits[i] * (1 << i)
        if i == 3:
            bit_shift = c_lower_bits
    send_byte(b, B_RANGE, (L.adapter.op_c_memory.prev_value[0] - c_lower_bits) * INV(64), 10, 0, is_real)
    for i in range(4):
        b.when(L.shift_u16[i]).assert_eq(L.c_bits[4] + L.c_bits[5] * 2 * not_word_sel, i)
        b.assert_bool(L.shift_u16[i])
    b.when(is_real).assert_eq(L.shift_u16[0] + L.shift_u16[1] + L.shift_u16[2] + L.shift_u16[3], 1)
    if left:
        b.assert_eq(L.v_01, (L.c_bits[0] + 1) * (L.c_bits[1] * 3 + 1))
        b.assert_eq(L.v_012, L.v_01 * (L.c_bits[2] * 15 + 1))
        b.assert_eq(L.v_0123, L.v_012 * (L.c_bits[3] * 255 + 1))
    else:
        b.assert_eq(L.v_01, (((1 - L.c_bits[0]) + 1) * 2) * ((1 - L.c_bits[1]) * 3 + 1))
        b.assert_eq(L.v_012, L.v_01 * ((1 - L.c_bits[2]) * 15 + 1))
        b.assert_eq(L.v_0123, L.v_012 * ((1 - L.c_bits[3]) * 255 + 1))
    return bit_shift


def shift_left_chip():
    b, c, _ = _chip("ShiftLeft", 65)
    L = S(("state", CPU_STATE), ("adapter", ALU_TYPE), ("a", 4), ("c_bits", 6), ("v_01", 1), ("v_012", 1), ("v_0123", 1),
          ("shift_u16", 4), ("lower_limb", 4), ("higher_limb", 4), ("limb_result", 4), ("sllw_msb", 1), ("is_sll", 1),
          ("is_sllw", 1), ("is_sllw_imm", 1))(c)
    is_real = L.is_sll + L.is_sllw
    for x in (is_real, L.is_sll, L.is_sllw):
        b.assert_bool(x)
    bit_shift = _shift_common(b, L, is_real, L.is_sll, left=True)
    bv = L.adapter.op_b_memory.prev_value
    for i in range(4):
        send_byte(b, B_RANGE, L.lower_limb[i], 16 - bit_shift, 0, is_real)
        send_byte(b, B_RANGE, L.higher_limb[i], bit_shift, 0, is_real)
        b.assert_eq(bv[i] * L.v_0123, L.higher_limb[i] * (1 << 16) + L.lower_limb[i] * L.v_0123)
    for i in range(4):
        r = L.lower_limb[i] * L.v_0123
        if i:
            r = r + L.higher_limb[i - 1]
        b.assert_eq(L.limb_result[i], r)
    for i in range(4):
        for j in range(4):
            if j < i:
                b.when(L.is_sll).when(L.shift_u16[i]).assert_zero(L.a[j])
            else:
                b.when(L.is_sll).when(L.shift_u16[i]).assert_eq(L.a[j], L.limb_result[j - i])
    for i in range(2):
        for j in range(2):
            if j < i:
                b.when(L.is_sllw).when(L.shift_u16[i]).assert_zero(L.a[j])
            else:
                b.when(L.is_sllw).when(L.shift_u16[i]).assert_eq(L.a[j], L.limb_result[j - i])
    for i in (2, 3):
        b.when(L.is_sllw).assert_eq(L.sllw_msb * 0xFFFF, L.a[i])
    eval_msb(b, L.a[1], L.sllw_msb, L.is_sllw)
    opcode = L.is_sll * OPC["SLL"] + L.is_sllw * OPC["SLLW"]
    b.assert_eq(L.is_sllw_imm, L.is_sllw * L.adapter.imm_c)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_alu_type(b, L.state, opcode, L.a, L.adapter, is_real, is_real)
    return _done(b, c)


def shift_right_chip():
    b, c, _ = _chip("ShiftRight", 69)
    L = S(("state", CPU_STATE), ("adapter", ALU_TYPE), ("a", 4), ("b_msb", 1), ("srw_msb", 1), ("c_bits", 6), ("sra_msb_v0123", 1),
          ("v_0123", 1), ("v_012", 1), ("v_01", 1), ("lower_limb", 4), ("higher_limb", 4), ("limb_result", 4), ("shift_u16", 4),
          ("is_srl", 1), ("is_sra", 1), ("is_srlw", 1), ("is_sraw", 1), ("is_w_imm", 1))(c)
    is_real = L.is_srl + L.is_sra + L.is_srlw + L.is_sraw
    for x in (L.is_srl, L.is_sra, L.is_srlw, L.is_sraw, is_real):
        b.assert_bool(x)
    is_word, not_word = L.is_srlw + L.is_sraw, L.is_srl + L.is_sra
    opcode = L.is_srl * OPC["SRL"] + L.is_sra * OPC["SRA"] + L.is_srlw * OPC["SRLW"] + L.is_sraw * OPC["SRAW"]
    b.assert_eq(L.is_w_imm, (L.is_srlw + L.is_sraw) * L.adapter.imm_c)
    bit_shift = _shift_common(b, L, is_real, not_word, left=False)
    bv = L.adapter.op_b_memory.prev_value
    for i in range(4):
        send_byte(b, B_RANGE, L.lower_limb[i], bit_shift, 0, is_real)
        send_byte(b, B_RANGE, L.higher_limb[i], 16 - bit_shift, 0, is_real)
        lhs = bv[i] * L.v_0123 if i < 2 else bv[i] * L.v_0123 * not_word
        b.assert_eq(lhs, L.higher_limb[i] * (1 << 16) + L.lower_limb[i] * L.v_0123)
    for i in range(4):
        r = L.higher_limb[i]
        if i != 3:
            r = r + L.lower_limb[i + 1] * L.v_0123
        b.assert_eq(L.limb_result[i], r)
    eval_msb(b, bv[3], L.b_msb, L.is_sra)
    eval_msb(b, bv[1], L.b_msb, L.is_sraw)
    b.when(L.is_srl + L.is_srlw).assert_zero(L.b_msb)
    b.assert_eq(L.sra_msb_v0123, L.b_msb * L.v_0123)
    eval_msb(b, L.a[1], L.srw_msb, is_word)
    b.when_not(is_word).assert_zero(L.srw_msb)
    fill = L.b_msb * (1 << 16) - L.sra_msb_v0123
    for i in range(4):
        for j in range(3 - i):
            b.when(not_word).when(L.shift_u16[i]).assert_eq(L.a[j], L.limb_result[i + j])
        b.when(not_word).when(L.shift_u16[i]).assert_eq(L.a[3 - i], L.limb_result[3] + fill)
        for j in range(4 - i, 4):
            b.when(not_word).when(L.shift_u16[i]).assert_eq(L.a[j], L.b_msb * 0xFFFF)
    b.when(is_word).when(L.shift_u16[0]).assert_eq(L.a[0], L.limb_result[0])
    b.when(is_word).when(L.shift_u16[0]).assert_eq(L.a[1], L.limb_result[1] + fill)
    b.when(is_word).when(L.shift_u16[1]).assert_eq(L.a[0], L.limb_result[1] + fill)
    b.when(is_word).when(L.shift_u16[1]).assert_eq(L.a[1], L.b_msb * 0xFFFF)
    for i in (2, 3):
        b.when(is_word).assert_eq(L.a[i], L.srw_msb * 0xFFFF)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_alu_type(b, L.state, opcode, L.a, L.adapter, is_real, is_real)
    return _done(b, c)


def utype_chip():
    b, c, _ = _chip("UType", 31)
    L = S(("state", CPU_STATE), ("adapter", J_TYPE), ("addend", 3), ("value", 4), ("is_auipc", 1), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    b.assert_bool(L.is_auipc)
    opcode = L.is_auipc * OPC["AUIPC"] + (1 - L.is_auipc) * OPC["LUI"]
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    addend = [L.addend[0], L.addend[1], L.addend[2], b.const(0)]
    expected = [b.if_else(L.is_auipc, x, 0) for x in (L.state.pc[0], L.state.pc[1], L.state.pc[2], b.const(0))]
    b.assert_word_eq(addend, expected)
    b.when_not(L.is_real).assert_zero(L.adapter.op_a_0)
    eval_add(b, addend, L.adapter.op_b_imm, L.value, L.is_real - L.adapter.op_a_0)
    eval_j_type(b, L.state, opcode, L.value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def memory_local_chip():
    b, c, _ = _chip("MemoryLocal", 20)
    L = S(("addr", 3), ("initial_clk_high", 1), ("final_clk_high", 1), ("initial_clk_low", 1), ("final_clk_low", 1),
          ("initial_value", 4), ("final_value", 4), ("initial_value_lower", 1), ("initial_value_upper", 1),
          ("final_value_lower", 1), ("final_value_upper", 1), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    cube = L.is_real * L.is_real * L.is_real
    b.assert_eq(cube, cube)                       # the reference's degree-3 filler (memory/local.rs:L266-L269)
    for tag, clk_high, clk_low, value, lower, upper, is_recv in (
            ("initial", L.initial_clk_high, L.initial_clk_low, L.initial_value, L.initial_value_lower, L.initial_value_upper, True),
            ("final", L.final_clk_high, L.final_clk_low, L.final_value, L.final_value_lower, L.final_value_upper, False)):
        b.assert_eq(value[2], lower + upper * (1 << 8))
        slice_range_check_u8(b, [lower, upper], L.is_real)
        slice_range_check_u16(b, value, L.is_real)
        msg = [clk_high, clk_low] + L.addr + value
        (b.receive if is_recv else b.send)(MEMORY, msg, L.is_real)
        b.send(GLOBAL, [clk_high, clk_low, L.addr[0], L.addr[1], L.addr[2], value[0] + lower * (1 << 16),
                        value[1] + upper * (1 << 16), value[3], 0 if is_recv else 1, 1 if is_recv else 0, MEMORY], L.is_real)
    return _done(b, c)


def memory_bump_chip():
    b, c, _ = _chip("MemoryBump", 15)
    L = S(("access", MEM_ACCESS), ("clk_32_48", 1), ("clk_24_32", 1), ("clk_16_24", 1), ("clk_0_16", 1), ("addr", 1), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    slice_range_check_u16(b, [L.clk_0_16, L.clk_32_48], L.is_real)
    slice_range_check_u8(b, [L.clk_16_24, L.clk_24_32], L.is_real)
    send_byte(b, B_LTU, 1, L.addr, 32, L.is_real)
    eval_memory_access(b, L.clk_24_32 + L.clk_32_48 * (1 << 8), L.clk_0_16 + L.clk_16_24 * (1 << 16), _reg_addr(b, L.addr), L.access,
                       L.access.prev_value, L.is_real)
    return _done(b, c)


def state_bump_chip():
    b, c, _ = _chip("StateBump", 14)
    L = S(("next_clk_32_48", 1), ("next_clk_24_32", 1), ("next_clk_16_24", 1), ("next_clk_0_16", 1), ("clk_high", 1), ("clk_low", 1),
          ("next_pc", 3), ("pc", 3), ("is_clk", 1), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    receive_state(b, L.clk_high, L.clk_low, L.pc, L.is_real)
    send_state(b, L.next_clk_24_32 + L.next_clk_32_48 * (1 << 8), L.next_clk_0_16 + L.next_clk_16_24 * (1 << 16), L.next_pc, L.is_real)
    send_byte(b, B_RANGE, (L.next_clk_0_16 - 1) * INV(8), 13, 0, L.is_real)
    send_byte(b, B_RANGE, L.next_clk_32_48, 16, 0, L.is_real)
    slice_range_check_u8(b, [L.next_clk_16_24, L.next_clk_24_32], L.is_real)
    b.assert_bool(L.is_clk)
    b.when(L.is_real).assert_eq(L.next_clk_24_32 + L.next_clk_32_48 * (1 << 8), L.clk_high + L.is_clk)
    b.when(L.is_real).assert_eq(L.next_clk_0_16 + L.next_clk_16_24 * (1 << 16) + L.is_clk * (1 << 24), L.clk_low)
    carry = b.const(0)
    for i in range(3):
        carry = (carry + L.pc[i] - L.next_pc[i]) * INV(1 << 16)
        b.assert_bool(carry)
    b.assert_zero(carry)
    slice_range_check_u16(b, L.next_pc, L.is_real)
    return _done(b, c)


def program_chip():
    b, c, cp = _chip("Program", 1, 16)
    Lp = S(("pc", 3), ("opcode", 1), ("op_a", 1), ("op_b", 4), ("op_c", 4), ("op_a_0", 1), ("imm_b", 1), ("imm_c", 1))(cp)
    mult = c.one("multiplicity")
    b.receive(PROGRAM, Lp.pc + instruction_values(Lp.opcode, Lp.op_a, Lp.op_b, Lp.op_c, Lp.op_a_0, Lp.imm_b, Lp.imm_c), mult)
    return _done(b, c, cp)


def byte_chip():
    b, c, cp = _chip("Byte", 6, 7)
    Lp = S(("b", 1), ("c", 1), ("and_", 1), ("or_", 1), ("xor", 1), ("ltu", 1), ("msb", 1))(cp)
    mult = c.arr(6, "multiplicities")
    # ByteOpcode::byte_table() order = AND, OR, XOR, U8Range, LTU, MSB (core/executor/src/opcode.rs)
    b.receive(BYTE, [B_AND, Lp.and_, Lp.b, Lp.c], mult[0])
    b.receive(BYTE, [B_OR, Lp.or_, Lp.b, Lp.c], mult[1])
    b.receive(BYTE, [B_XOR, Lp.xor, Lp.b, Lp.c], mult[2])
    b.receive(BYTE, [B_U8RANGE, 0, Lp.b, Lp.c], mult[3])
    b.receive(BYTE, [B_LTU, Lp.ltu, Lp.b, Lp.c], mult[4])
    b.receive(BYTE, [B_MSB, Lp.msb, Lp.b, 0], mult[5])
    return _done(b, c, cp)


def range_chip():
    b, c, cp = _chip("Range", 1, 2)
    Lp = S(("a", 1), ("bits", 1))(cp)
    mult = c.one("multiplicity")
    b.receive(BYTE, [B_RANGE, Lp.a, Lp.bits, 0], mult)
    return _done(b, c, cp)


def addw_chip():
    b, c, _ = _chip("Addw", 36)
    L = S(("state", CPU_STATE), ("adapter", ALU_TYPE), ("value", 2), ("msb", 1), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_addw(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_memory.prev_value, L.value, L.msb, L.is_real)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    word = [L.value[0], L.value[1], L.msb * 0xFFFF, L.msb * 0xFFFF]
    eval_alu_type(b, L.state, OPC["ADDW"], word, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def subw_chip():
    b, c, _ = _chip("Subw", 32)
    L = S(("state", CPU_STATE), ("adapter", R_TYPE), ("value", 2), ("msb", 1), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_subw(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_memory.prev_value, L.value, L.msb, L.is_real)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    word = [L.value[0], L.value[1], L.msb * 0xFFFF, L.msb * 0xFFFF]
    eval_r_type(b, L.state, OPC["SUBW"], word, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def _mem_clk(st):
    return st.clk_high, clk_low_of(st) + POS_MEMORY


def load_double_chip():                                                    # memory/instructions/load/load_double.rs:L190-L299
    b, c, _ = _chip("LoadDouble", 39)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, 0, 0, 0, L.is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.memory_access.prev_value, L.is_real)
    b.assert_zero(L.adapter.op_a_0)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_i_type(b, L.state, OPC["LD"], L.memory_access.prev_value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def load_byte_chip():                                                      # load_byte.rs:L240-L408
    b, c, _ = _chip("LoadByte", 47)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("offset_bit", 3),
          ("selected_limb", 1), ("selected_limb_low_byte", 1), ("selected_byte", 1), ("msb", 1), ("is_lb", 1), ("is_lbu", 1))(c)
    opcode = L.is_lb * OPC["LB"] + L.is_lbu * OPC["LBU"]
    is_real = L.is_lb + L.is_lbu
    for x in (L.is_lb, L.is_lbu, is_real):
        b.assert_bool(x)
    ob = L.offset_bit
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, ob[0], ob[1], ob[2], is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.memory_access.prev_value, is_real)
    b.assert_zero(L.adapter.op_a_0)
    pv = L.memory_access.prev_value
    b.when_not(ob[1]).when_not(ob[2]).assert_eq(L.selected_limb, pv[0])
    b.when(ob[1]).when_not(ob[2]).assert_eq(L.selected_limb, pv[1])
    b.when_not(ob[1]).when(ob[2]).assert_eq(L.selected_limb, pv[2])
    b.when(ob[1]).when(ob[2]).assert_eq(L.selected_limb, pv[3])
    byte0 = L.selected_limb_low_byte
    byte1 = (L.selected_limb - byte0) * INV(1 << 8)
    slice_range_check_u8(b, [byte0, byte1], is_real)
    b.assert_eq(L.selected_byte, ob[0] * byte1 + (1 - ob[0]) * byte0)
    b.when(L.is_lbu).assert_zero(L.msb)
    send_byte(b, B_MSB, L.msb, L.selected_byte, 0, L.is_lb)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    word = [L.selected_byte + ((1 << 16) - (1 << 8)) * L.msb, 0xFFFF * L.msb, 0xFFFF * L.msb, 0xFFFF * L.msb]
    eval_i_type(b, L.state, opcode, word, L.adapter, is_real, is_real)
    return _done(b, c)


def load_half_chip():                                                      # load_half.rs
    b, c, _ = _chip("LoadHalf", 44)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("offset_bit", 2),
          ("selected_half", 1), ("msb", 1), ("is_lh", 1), ("is_lhu", 1))(c)
    opcode = L.is_lh * OPC["LH"] + L.is_lhu * OPC["LHU"]
    is_real = L.is_lh + L.is_lhu
    for x in (L.is_lh, L.is_lhu, is_real):
        b.assert_bool(x)
    ob = L.offset_bit
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, 0, ob[0], ob[1], is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.memory_access.prev_value, is_real)
    b.assert_zero(L.adapter.op_a_0)
    pv = L.memory_access.prev_value
    b.when_not(ob[0]).when_not(ob[1]).assert_eq(L.selected_half, pv[0])
    b.when(ob[0]).when_not(ob[1]).assert_eq(L.selected_half, pv[1])
    b.when_not(ob[0]).when(ob[1]).assert_eq(L.selected_half, pv[2])
    b.when(ob[0]).when(ob[1]).assert_eq(L.selected_half, pv[3])
    b.when(L.is_lhu).assert_zero(L.msb)
    eval_msb(b, L.selected_half, L.msb, L.is_lh)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    word = [L.selected_half, 0xFFFF * L.msb, 0xFFFF * L.msb, 0xFFFF * L.msb]
    eval_i_type(b, L.state, opcode, word, L.adapter, is_real, is_real)
    return _done(b, c)


def load_word_chip():                                                      # load_word.rs
    b, c, _ = _chip("LoadWord", 44)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("offset_bit", 1),
          ("selected_word", 2), ("msb", 1), ("is_lw", 1), ("is_lwu", 1))(c)
    opcode = L.is_lw * OPC["LW"] + L.is_lwu * OPC["LWU"]
    is_real = L.is_lw + L.is_lwu
    for x in (L.is_lw, L.is_lwu, is_real):
        b.assert_bool(x)
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, 0, 0, L.offset_bit, is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.memory_access.prev_value, is_real)
    b.assert_zero(L.adapter.op_a_0)
    pv = L.memory_access.prev_value
    b.when_not(L.offset_bit).assert_eq(L.selected_word[0], pv[0])
    b.when_not(L.offset_bit).assert_eq(L.selected_word[1], pv[1])
    b.when(L.offset_bit).assert_eq(L.selected_word[0], pv[2])
    b.when(L.offset_bit).assert_eq(L.selected_word[1], pv[3])
    eval_msb(b, L.selected_word[1], L.msb, L.is_lw)
    b.when_not(L.is_lw).assert_zero(L.msb)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    word = [L.selected_word[0], L.selected_word[1], 0xFFFF * L.msb, 0xFFFF * L.msb]
    eval_i_type(b, L.state, opcode, word, L.adapter, is_real, is_real)
    return _done(b, c)


def load_x0_chip():                                                        # load/load_x0.rs:L220-L388
    """Loads whose destination is x0: the access to memory (and its alignment / address checks) happens, the value goes nowhere."""
    b, c, _ = _chip("LoadX0", 48)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("offset_bit", 3),
          ("is_lb", 1), ("is_lbu", 1), ("is_lh", 1), ("is_lhu", 1), ("is_lw", 1), ("is_lwu", 1), ("is_ld", 1))(c)
    sel = [("LB", L.is_lb), ("LBU", L.is_lbu), ("LH", L.is_lh), ("LHU", L.is_lhu), ("LW", L.is_lw), ("LWU", L.is_lwu), ("LD", L.is_ld)]
    opcode = sum((f * OPC[n] for n, f in sel[1:]), sel[0][1] * OPC[sel[0][0]])
    is_real = L.is_lb + L.is_lbu + L.is_lh + L.is_lhu + L.is_lw + L.is_lwu + L.is_ld
    for _, f in sel:
        b.assert_bool(f)
    b.assert_bool(is_real)
    ob = L.offset_bit
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, ob[0], ob[1], ob[2], is_real, L.address)
    b.when(L.is_ld).assert_zero(ob[2])
    b.when(L.is_lw + L.is_lwu + L.is_ld).assert_zero(ob[1])
    b.when(L.is_lh + L.is_lhu + L.is_lw + L.is_lwu + L.is_ld).assert_zero(ob[0])
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.memory_access.prev_value, is_real)
    b.when(is_real).assert_one(L.adapter.op_a_0)
    b.when_not(is_real).assert_zero(L.adapter.op_a_0)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, is_real)
    eval_i_type(b, L.state, opcode, L.adapter.op_a_memory.prev_value, L.adapter, is_real, is_real)      # ITypeReaderImmutable
    return _done(b, c)


def store_double_chip():                                                   # store/store_double.rs
    b, c, _ = _chip("StoreDouble", 39)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, 0, 0, 0, L.is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.adapter.op_a_memory.prev_value, L.is_real)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_i_type(b, L.state, OPC["SD"], L.adapter.op_a_memory.prev_value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def store_word_chip():                                                     # store/store_word.rs
    b, c, _ = _chip("StoreWord", 44)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("offset_bit", 1),
          ("store_value", 4), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, 0, 0, L.offset_bit, L.is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.store_value, L.is_real)
    sl, pv, ob = L.adapter.op_a_memory.prev_value, L.memory_access.prev_value, L.offset_bit
    b.assert_eq(L.store_value[0], pv[0] + (sl[0] - pv[0]) * (1 - ob))
    b.assert_eq(L.store_value[1], pv[1] + (sl[1] - pv[1]) * (1 - ob))
    b.assert_eq(L.store_value[2], pv[2] + (sl[0] - pv[2]) * ob)
    b.assert_eq(L.store_value[3], pv[3] + (sl[1] - pv[3]) * ob)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_i_type(b, L.state, OPC["SW"], L.adapter.op_a_memory.prev_value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def store_half_chip():                                                     # store/store_half.rs
    b, c, _ = _chip("StoreHalf", 45)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("offset_bit", 2),
          ("store_value", 4), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    ob = L.offset_bit
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, 0, ob[0], ob[1], L.is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.store_value, L.is_real)
    sl, pv = L.adapter.op_a_memory.prev_value[0], L.memory_access.prev_value
    b.assert_eq(L.store_value[0], pv[0] + (sl - pv[0]) * (1 - ob[0]) * (1 - ob[1]))
    b.assert_eq(L.store_value[1], pv[1] + (sl - pv[1]) * ob[0] * (1 - ob[1]))
    b.assert_eq(L.store_value[2], pv[2] + (sl - pv[2]) * (1 - ob[0]) * ob[1])
    b.assert_eq(L.store_value[3], pv[3] + (sl - pv[3]) * ob[0] * ob[1])
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_i_type(b, L.state, OPC["SH"], L.adapter.op_a_memory.prev_value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def store_byte_chip():                                                     # store/store_byte.rs
    b, c, _ = _chip("StoreByte", 50)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("address", ADDRESS_OP), ("memory_access", MEM_ACCESS), ("offset_bit", 3),
          ("mem_limb", 1), ("mem_limb_low_byte", 1), ("register_low_byte", 1), ("increment", 1), ("store_value", 4), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    ob = L.offset_bit
    aligned = eval_address(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, ob[0], ob[1], ob[2], L.is_real, L.address)
    eval_memory_access(b, *_mem_clk(L.state), aligned, L.memory_access, L.store_value, L.is_real)
    pv = L.memory_access.prev_value
    b.when_not(ob[1]).when_not(ob[2]).assert_eq(L.mem_limb, pv[0])
    b.when(ob[1]).when_not(ob[2]).assert_eq(L.mem_limb, pv[1])
    b.when_not(ob[1]).when(ob[2]).assert_eq(L.mem_limb, pv[2])
    b.when(ob[1]).when(ob[2]).assert_eq(L.mem_limb, pv[3])
    byte0 = L.register_low_byte
    byte1 = (L.adapter.op_a_memory.prev_value[0] - byte0) * INV(1 << 8)
    slice_range_check_u8(b, [byte0, byte1], L.is_real)
    byte0 = L.mem_limb_low_byte
    byte1 = (L.mem_limb - byte0) * INV(1 << 8)
    slice_range_check_u8(b, [byte0, byte1], L.is_real)
    b.assert_eq(L.increment, (L.register_low_byte - L.mem_limb_low_byte) * (1 - ob[0])
                + (1 << 8) * (L.register_low_byte - byte1) * ob[0])
    b.assert_eq(L.store_value[0], L.increment * (1 - ob[1]) * (1 - ob[2]) + pv[0])
    b.assert_eq(L.store_value[1], L.increment * ob[1] * (1 - ob[2]) + pv[1])
    b.assert_eq(L.store_value[2], L.increment * (1 - ob[1]) * ob[2] + pv[2])
    b.assert_eq(L.store_value[3], L.increment * ob[1] * ob[2] + pv[3])
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    eval_i_type(b, L.state, OPC["SB"], L.adapter.op_a_memory.prev_value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def branch_chip():                                                         # control_flow/branch/air.rs:L31-L214
    b, c, _ = _chip("Branch", 45)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("next_pc", 3), ("is_beq", 1), ("is_bne", 1), ("is_blt", 1), ("is_bge", 1),
          ("is_bltu", 1), ("is_bgeu", 1), ("is_branching", 1), ("cmp", LT_SIGNED))(c)
    sel = (L.is_beq, L.is_bne, L.is_blt, L.is_bge, L.is_bltu, L.is_bgeu)
    for x in sel:
        b.assert_bool(x)
    is_real = L.is_beq + L.is_bne + L.is_blt + L.is_bge + L.is_bltu + L.is_bgeu
    b.assert_bool(is_real)
    opcode = (L.is_beq * OPC["BEQ"] + L.is_bne * OPC["BNE"] + L.is_blt * OPC["BLT"] + L.is_bge * OPC["BGE"]
              + L.is_bltu * OPC["BLTU"] + L.is_bgeu * OPC["BGEU"])
    eval_cpu_state(b, L.state, L.next_pc, CLK_INC, is_real)
    eval_i_type(b, L.state, opcode, L.adapter.op_a_memory.prev_value, L.adapter, is_real, is_real)
    eval_lt_signed(b, L.adapter.op_a_memory.prev_value, L.adapter.op_b_memory.prev_value, L.cmp, L.is_blt + L.is_bge, is_real)
    f = L.cmp.result.u16_flags
    is_eq = 1 - (f[0] + f[1] + f[2] + f[3])
    lt = L.cmp.result.bit
    branching = L.is_beq * is_eq
    branching = branching + L.is_bne * (1 - is_eq)
    branching = branching + (L.is_bge + L.is_bgeu) * (1 - lt)
    branching = branching + (L.is_blt + L.is_bltu) * lt
    b.assert_bool(L.is_branching)
    b.when(is_real).assert_eq(L.is_branching, branching)
    imm = L.adapter.op_c_imm
    carry = b.const(0)
    for i in range(4):
        pc = L.state.pc[i] if i < 3 else b.const(0)
        npc = L.next_pc[i] if i < 3 else b.const(0)
        carry = (carry + pc + imm[i] - npc) * INV(1 << 16)
        b.when(L.is_branching).assert_bool(carry)
    carry = b.const(0)
    for i in range(4):
        pc = L.state.pc[i] if i < 3 else b.const(0)
        npc = L.next_pc[i] if i < 3 else b.const(0)
        carry = (carry + pc + (PC_INC if i == 0 else 0) - npc) * INV(1 << 16)
        b.when(is_real - L.is_branching).assert_bool(carry)
    send_byte(b, B_RANGE, L.next_pc[0] * INV(4), 14, 0, is_real)
    slice_range_check_u16(b, L.next_pc[1:3], is_real)
    return _done(b, c)


def _pc_word(b, st):
    return [st.pc[0], st.pc[1], st.pc[2], b.const(0)]


def jal_chip():                                                            # control_flow/jal/air.rs
    b, c, _ = _chip("Jal", 31)
    L = S(("state", CPU_STATE), ("adapter", J_TYPE), ("next_pc", 4), ("op_a_value", 4), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    eval_add(b, _pc_word(b, L.state), L.adapter.op_b_imm, L.next_pc, L.is_real)
    b.assert_zero(L.next_pc[3])
    send_byte(b, B_RANGE, L.next_pc[0] * INV(4), 14, 0, L.is_real)
    eval_cpu_state(b, L.state, L.next_pc[:3], CLK_INC, L.is_real)
    b.when_not(L.is_real).assert_zero(L.adapter.op_a_0)
    eval_add(b, _pc_word(b, L.state), [4, 0, 0, 0], L.op_a_value, L.is_real - L.adapter.op_a_0)
    b.assert_zero(L.op_a_value[3])
    for i in range(3):
        b.when(L.adapter.op_a_0).assert_zero(L.op_a_value[i])
    eval_j_type(b, L.state, OPC["JAL"], L.op_a_value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


def jalr_chip():                                                           # control_flow/jalr/air.rs
    b, c, _ = _chip("Jalr", 35)
    L = S(("state", CPU_STATE), ("adapter", I_TYPE), ("is_real", 1), ("next_pc", 4), ("op_a_value", 4), ("lsb", 1))(c)
    b.assert_bool(L.is_real)
    eval_add(b, L.adapter.op_b_memory.prev_value, L.adapter.op_c_imm, L.next_pc, L.is_real)
    b.assert_zero(L.next_pc[3])
    b.assert_bool(L.lsb)
    send_byte(b, B_RANGE, (L.next_pc[0] - L.lsb) * INV(4), 14, 0, L.is_real)
    eval_cpu_state(b, L.state, [L.next_pc[0] - L.lsb, L.next_pc[1], L.next_pc[2]], CLK_INC, L.is_real)
    eval_i_type(b, L.state, OPC["JALR"], L.op_a_value, L.adapter, L.is_real, L.is_real)
    b.when_not(L.is_real).assert_zero(L.adapter.op_a_0)
    eval_add(b, _pc_word(b, L.state), [4, 0, 0, 0], L.op_a_value, L.is_real - L.adapter.op_a_0)
    b.assert_zero(L.op_a_value[3])
    for i in range(3):
        b.when(L.adapter.op_a_0).assert_zero(L.op_a_value[i])
    return _done(b, c)



# ---- the Global chip: septic-curve digest of every global interaction (global/mod.rs, operations/global_interaction.rs,
# operations/global_accumulation.rs; F_p^7 = F_p[z] / (z^7 - 3 z - 5), curve y^2 = x^3 + 45 x + 41 z^3: hypercube/src/septic_*.rs)
CURVE_CUMULATIVE_SUM_START = ([0x1414213, 0x5623730, 0x9504880, 0x1688724, 0x2096980, 0x7856967, 0x1875376],
                              [2020310104, 1513506566, 1843922297, 2003644209, 805967281, 1882435203, 1623804682])   # septic_digest.rs:L10-L16
CURVE_WITNESS_DUMMY_POINT = ([0x2718281 + (1 << 24), 0x8284590, 0x4523536, 0x0287471, 0x3526624, 0x9775724, 0x7093699],
                             [1250555984, 1592495468, 656721246, 420301347, 2125819749, 819876460, 17687681])         # septic_curve.rs:L23-L28


def septic_mul(b, x, y):                                                   # septic_extension.rs:L307-L325
    res = [b.const(0)] * 13
    for i in range(7):
        for j in range(7):
            res[i + j] = res[i + j] + x[i] * y[j]
    ret = list(res[:7])
    for i in range(7, 13):
        ret[i - 7] = ret[i - 7] + res[i] * 5
        ret[i - 6] = ret[i - 6] + res[i] * 3
    return ret


def septic_add(x, y):
    return [a + c for a, c in zip(x, y)]


def septic_sub(x, y):
    return [a - c for a, c in zip(x, y)]


def curve_formula(b, x):                                                   # septic_curve.rs:L101-L113
    cube = septic_mul(b, septic_mul(b, x, x), x)
    out = [c + xi * 45 for c, xi in zip(cube, x)]
    out[3] = out[3] + 41
    return out


GLOBAL_INTERACTION = S(("x_coordinate", 7), ("y_coordinate", 7), ("permutation", 179), ("offset", 1), ("y6_byte_decomp", 4))
GLOBAL_ACCUMULATION = S(("initial_digest_x", 7), ("initial_digest_y", 7), ("cumulative_sum_x", 7), ("cumulative_sum_y", 7))


def septic_product_coeffs(b, xf, yf, square=False):
    """The seven coefficients of x y in F_p[z] / (z^7 - 3 z - 5), as independent expression trees emitted COEFFICIENT by
    coefficient: ret[k] = T_k + 5 T_(k+7) + 3 T_(k+6) with T_s = sum_{i+j=s} x_i y_j, so only one group sum T_s is shared —
    between coefficients k and k + 1 — and nothing else stays live (the reference's `Mul` accumulates all 13 group sums at
    once: same polynomials, 13 + 14 live values). xf(i) / yf(i) BUILD operand i afresh at every use (cheap affine forms are
    recomputed rather than kept in a register); square=True uses the symmetry of x x (28 products instead of 49)."""
    T = {}

    def group(s_):
        if s_ not in T:
            acc = None
            for i in range(7):
                j = s_ - i
                if not 0 <= j < 7:
                    continue
                if square:
                    if i > j:
                        continue
                    t = xf(i) * xf(j)
                    if i < j:
                        t = t * 2
                else:
                    t = xf(i) * yf(j)
                acc = t if acc is None else acc + t
            T[s_] = acc
        return T[s_]
    out = []
    for k in range(7):
        r = group(k)
        if k + 7 <= 12:
            r = r + group(k + 7) * 5
        if k >= 1:
            r = r + group(k + 6) * 3
        out.append(r)
    return out


def global_chip(form=None):
    """form="reference": septic products accumulated like the reference's `SepticExtension::mul` (13 group sums at once);
    "coefficient-major" (default): the same constraint polynomials emitted coefficient by coefficient with cheap operands
    recomputed — 12 live values instead of 41, so the zerocheck interpreter runs the chip four times as wide (DESIGN.md §7.2;
    tests/test_riscv_machine.py checks the two forms agree on random rows)."""
    import os
    from .recursion import P2_EXT, P2_OUT, poseidon2_permutation_constraints
    form = form or os.environ.get("SP1_GLOBAL_FORM", "coefficient-major")
    b, c, _ = _chip("Global", 241)
    L = S(("message", 8), ("kind", 1), ("message_0_16bit_limb", 1), ("message_0_8bit_limb", 1), ("interaction", GLOBAL_INTERACTION),
          ("is_real", 1), ("is_receive", 1), ("is_send", 1), ("index", 1), ("accumulation", GLOBAL_ACCUMULATION))(c)
    I, A = L.interaction, L.accumulation
    b.assert_bool(L.is_real)
    b.receive(GLOBAL, L.message + [L.is_send, L.is_receive, L.kind], L.is_real)
    # GlobalInteractionOperation::eval_single_digest (operations/global_interaction.rs:L106-L237)
    b.assert_bool(L.is_real)
    b.when(L.is_real).assert_eq(L.is_receive + L.is_send, 1)
    b.assert_bool(L.is_receive)
    b.assert_bool(L.is_send)
    send_byte(b, B_U8RANGE, 0, 0, I.offset, L.is_real)
    b.when(L.is_real).assert_eq(L.message[0], L.message_0_16bit_limb + L.message_0_8bit_limb * (1 << 16))
    slice_range_check_u16(b, [L.message_0_16bit_limb, L.message[7]], L.is_real)
    slice_range_check_u8(b, [L.message_0_8bit_limb], L.is_real)
    send_byte(b, B_RANGE, L.kind, 6, 0, L.is_real)
    m_trial = [L.message[0] + (1 << 24) * L.kind] + L.message[1:7] + [L.message[7] + (1 << 16) * I.offset] + [b.const(0)] * 8
    perm = I.permutation
    for i in range(16):
        b.when(L.is_real).assert_eq(perm[P2_EXT(0, i)], m_trial[i])
    base = c.names["interaction.permutation"]
    poseidon2_permutation_constraints(b.air, base)             # (switches the AirProgram's hash-consing off)
    xc, yc = c.names["interaction.x_coordinate"], c.names["interaction.y_coordinate"]
    ac = {nm: c.names["accumulation." + nm] for nm in ("initial_digest_x", "initial_digest_y", "cumulative_sum_x", "cumulative_sum_y")}
    if form == "reference":
        b.air._seen = {}
        for i in range(7):
            b.when(L.is_real).assert_eq(I.x_coordinate[i], perm[P2_OUT(i)])
        x, y = I.x_coordinate, I.y_coordinate
        b.assert_all_eq(septic_mul(b, y, y), curve_formula(b, x))
    else:
        from .rv_builder import Sym, _MAIN
        fresh = lambda col: Sym(b, _MAIN, col, None, ({("main", col): 1}, 0))     # a new load at every use
        if form == "coefficient-major-shared-xy":                                # x, y loaded once (14 registers), the rest afresh
            _xy = {col: b.main(col) for col in list(range(xc, xc + 7)) + list(range(yc, yc + 7))}
            _f0 = fresh
            fresh = lambda col: _xy[col] if col in _xy else _f0(col)
        for i in range(7):
            b.when(L.is_real).assert_eq(fresh(xc + i), fresh(base + P2_OUT(i)))
        b.air.hint_septic_curve(xc)
        y2 = septic_product_coeffs(b, lambda i: fresh(yc + i), None, square=True)
        x2 = septic_product_coeffs(b, lambda i: fresh(xc + i), None, square=True)
        x3 = septic_product_coeffs(b, lambda i: x2[i], lambda j: fresh(xc + j))
        for k in range(7):
            rhs = x3[k] + fresh(xc + k) * 45
            if k == 3:
                rhs = rhs + 41
            b.assert_eq(y2[k], rhs)
        y = [fresh(yc + i) for i in range(7)]
    y6_value = b.const(0)
    for i in range(3):
        y6_value = y6_value + I.y6_byte_decomp[i] * (1 << (8 * i))
        send_byte(b, B_U8RANGE, 0, 0, I.y6_byte_decomp[i], L.is_real)
    y6_value = y6_value + I.y6_byte_decomp[3] * (1 << 24)
    send_byte(b, B_LTU, 1, I.y6_byte_decomp[3], 63, L.is_real)
    b.when(L.is_receive).assert_eq(y[6], 1 + y6_value)
    b.when(L.is_send).assert_zero(y[6] + 1 + y6_value)
    # GlobalAccumulationOperation::eval_accumulation (operations/global_accumulation.rs:L56-L146)
    b.assert_bool(L.is_real)
    b.receive(GLOBAL_ACC, [L.index] + A.initial_digest_x + A.initial_digest_y, L.is_real)
    p1x, p1y, p3x, p3y = A.initial_digest_x, A.initial_digest_y, A.cumulative_sum_x, A.cumulative_sum_y
    if form == "reference":
        dx, dy = septic_sub(x, p1x), septic_sub(y, p1y)
        checker_x = septic_sub(septic_mul(b, septic_add(septic_add(p1x, x), p3x), septic_mul(b, dx, dx)), septic_mul(b, dy, dy))
        checker_y = septic_sub(septic_mul(b, septic_add(p1y, p3y), dx), septic_mul(b, dy, septic_sub(p1x, p3x)))
        b.assert_all_eq(checker_x, [0] * 7)
        b.when(L.is_real).assert_all_eq(checker_y, [0] * 7)
    else:
        b.air.hint_septic_sum(xc, ac["initial_digest_x"], c.names["is_real"])
        dxf = lambda i: fresh(xc + i) - fresh(ac["initial_digest_x"] + i)
        dyf = lambda i: fresh(yc + i) - fresh(ac["initial_digest_y"] + i)
        sxf = lambda i: (fresh(ac["initial_digest_x"] + i) + fresh(xc + i)) + fresh(ac["cumulative_sum_x"] + i)
        dx2 = septic_product_coeffs(b, dxf, None, square=True)                  # 7 values kept across the next product
        sd = septic_product_coeffs(b, sxf, lambda j: dx2[j])
        dy2 = septic_product_coeffs(b, dyf, None, square=True)
        for k in range(7):
            b.assert_eq(sd[k] - dy2[k], 0)
        syf = lambda i: fresh(ac["initial_digest_y"] + i) + fresh(ac["cumulative_sum_y"] + i)
        pxf = lambda i: fresh(ac["initial_digest_x"] + i) - fresh(ac["cumulative_sum_x"] + i)
        u = septic_product_coeffs(b, syf, dxf)
        v = septic_product_coeffs(b, dyf, pxf)
        for k in range(7):
            b.when(L.is_real).assert_eq(u[k] - v[k], 0)
    b.send(GLOBAL_ACC, [L.index + 1] + p3x + p3y, L.is_real)
    return _done(b, c)


CHIPS = {
    "Add": add_chip, "Addi": addi_chip, "Sub": sub_chip, "Bitwise": bitwise_chip, "Lt": lt_chip, "Mul": mul_chip,
    "ShiftLeft": shift_left_chip, "ShiftRight": shift_right_chip, "UType": utype_chip, "MemoryLocal": memory_local_chip,
    "Addw": addw_chip, "Subw": subw_chip, "LoadByte": load_byte_chip, "LoadHalf": load_half_chip, "LoadWord": load_word_chip,
    "LoadDouble": load_double_chip, "LoadX0": load_x0_chip, "StoreByte": store_byte_chip, "StoreHalf": store_half_chip, "StoreWord": store_word_chip,
    "StoreDouble": store_double_chip, "Branch": branch_chip, "Jal": jal_chip, "Jalr": jalr_chip, "MemoryBump": memory_bump_chip, "StateBump": state_bump_chip, "Program": program_chip, "Byte": byte_chip, "Range": range_chip, "Global": global_chip,
}

_CACHE = {}


def chip(name):
    """(AirProgram, InteractionProgram) of a transcribed chip; `air.layout` maps dotted column names to indices."""
    if name not in _CACHE:
        if name in CHIPS:
            _CACHE[name] = CHIPS[name]()
        else:                                     # DivRem, the syscall chips, global memory init / finalize, Keccak (riscv_more.py)
            from .riscv_more import MORE_CHIPS
            _CACHE[name] = MORE_CHIPS[name]()
    return _CACHE[name]


def stats(name):
    air, it = chip(name)
    return {"columns": air.main_width + air.prep_width, "constraints": air.num_constraints, "interactions": it.num_interactions,
            "instructions": len(air.instrs)}
