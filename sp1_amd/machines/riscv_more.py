"""More chips of SP1 v6's rv64im machine as DATA: what real programs' shards contain beyond the 30 chips of
riscv.py — DivRem, the syscall chips, global memory initialisation / finalisation, and one precompile with its controller.

    chip                  width  constraints  reference eval
    AluX0                    34       17      alu/alu_x0.rs:L236-L340
    DivRem                  246      348      alu/divrem/mod.rs:L597-L1313
    SyscallCore              10        2      syscall/chip.rs:L299-L420   (shard_kind = Core)
    SyscallPrecompile        10        2      syscall/chip.rs:L299-L420   (shard_kind = Precompile)
    SyscallInstrs            65       93      syscall/instructions/air.rs:L28-L595
    MemoryGlobalInit         30       31      memory/global.rs:L307-L474  (kind = Initialize)
    MemoryGlobalFinalize     30       31      memory/global.rs:L307-L474  (kind = Finalize)
    KeccakPermute          2640     2859      syscall/precompiles/keccak256/air.rs:L29-L200
    KeccakPermuteControl    634      331      syscall/precompiles/keccak256/controller.rs:L243-L398
    Poseidon2               348      497      syscall/precompiles/poseidon2/air.rs:L424-L607

Same method and the same pins as riscv.py: every `eval` (Supervisor mode, `mprotect` off) transcribed operation by operation in
the reference's call order against the recording builder; column counts == rv64im_costs.json, `assert_zero` counts ==
rv64im_complexity.json (tests/test_riscv_more.py), and semantics through executed traces (riscv_more_trace.py): every constraint
vanishes on every row, every bus balances.

NOT pinned by anything in the reference tree: the ORDER of the fields inside `KeccakCols` — that struct lives in the un-vendored
`p3-keccak-air` dependency (slop/crates/keccak-air/src/lib.rs re-exports it). Its size (2633 = 2640 - 7) and every use the
reference makes of it (keccak256/air.rs) are pinned; the field order below is the published Plonky3 one. A different order
permutes columns, not polynomials.
"""
import types

from ..air import P
from .rv_builder import Builder, Sym
from .riscv import (ADDRESS_OP, ALU_TYPE, B_AND, B_LTU, B_RANGE, B_U8RANGE, B_XOR, BYTE, CLK_INC, CPU_STATE, GLOBAL, INV, LT_UNSIGNED, MEM_ACCESS, MEMORY, MUL_OP, OPC,
                    PC_INC, R_TYPE, S, SYSCALL, U16_TO_U8, _chip, _done, clk_low_of, eval_add, eval_addr_add, eval_alu_type, eval_compare_u16, eval_cpu_state,
                    eval_lt_unsigned, eval_memory_access, eval_msb, eval_mul, eval_r_type, next_pc_inc, send_byte, slice_range_check_u16,
                    slice_range_check_u8, u16_to_u8_safe)

KECCAK, MEMORY_GLOBAL_INIT_CONTROL, MEMORY_GLOBAL_FINALIZE_CONTROL = 12, 14, 15         # hypercube/src/lookup/interaction.rs:L53-L62
# SyscallCode (core/executor/src/syscall_code.rs:L48-L105): byte 0 = syscall id, byte 1 = "has its own table"
SYS_HALT, SYS_ENTER_UNCONSTRAINED, SYS_COMMIT, SYS_COMMIT_DEFERRED_PROOFS, SYS_HINT_LEN, SYS_KECCAK_PERMUTE = 0x00, 0x03, 0x10, 0x1A, 0xF0, 0x09
SYS_POSEIDON2 = 0x33
HALT_PC = 1                                                                              # core/executor/src/lib.rs:L100
U16_MAX = 0xFFFF

# PublicValues<[T; 4], [T; 3], [T; 4], T> (hypercube/src/air/public_values.rs:L33-L168, `mprotect` off): word offsets
PV_COMMITTED_VALUE_DIGEST, PV_DEFERRED_PROOFS_DIGEST, PV_EXIT_CODE = 32, 72, 87
PV_COMMIT_SYSCALL, PV_COMMIT_DEFERRED_SYSCALL, PV_NUM_ELTS = 145, 147, 160

IS_ZERO = S(("inverse", 1), ("result", 1))                                               # operations/is_zero.rs:L27-L34
IS_ZERO_WORD = S(("is_zero_limb", lambda c, p: [IS_ZERO(c, p + "%d." % i) for i in range(4)]), ("is_zero_first_half", 1),
                 ("is_zero_second_half", 1), ("result", 1))                              # operations/is_zero_word.rs:L31-L43
SYSCALL_ADDR = S(("addr", 3), ("top_two_limb_min", 1), ("top_two_limb_max", IS_ZERO))    # operations/syscall_addr.rs:L14-L24
ADD_OP = S(("value", 4),)                                                                # operations/add.rs:L27-L31
ADDR_ADD_OP = S(("value", 3),)                                                           # operations/addrs_add.rs:L24-L28


def eval_is_zero(b, a, cols, is_real):                                                   # operations/is_zero.rs:L59-L83
    is_zero = 1 - cols.inverse * a
    b.when(is_real).assert_eq(is_zero, cols.result)
    b.when(is_real).assert_bool(cols.result)
    b.when(is_real).when(cols.result).assert_zero(a)


def eval_is_zero_word(b, a, cols, is_real):                                              # operations/is_zero_word.rs:L62-L101
    for i in range(4):
        eval_is_zero(b, a[i], cols.is_zero_limb[i], is_real)
    b.assert_bool(is_real)
    b.assert_bool(cols.result)
    b.assert_eq(cols.is_zero_first_half, cols.is_zero_limb[0].result * cols.is_zero_limb[1].result)
    b.assert_eq(cols.is_zero_second_half, cols.is_zero_limb[2].result * cols.is_zero_limb[3].result)
    b.when(is_real).assert_eq(cols.result, cols.is_zero_first_half * cols.is_zero_second_half)


def eval_is_equal_word(b, x, y, cols, is_real):                                          # operations/is_equal_word.rs:L55-L80
    b.assert_bool(is_real)
    eval_is_zero_word(b, [b._s(x[i]) - b._s(y[i]) for i in range(4)], cols, is_real)


def word_of_u64(v):
    return [(v >> (16 * i)) & 0xFFFF for i in range(4)]


def send_syscall(b, clk_high, clk_low, syscall_id, arg1, arg2, mult, receive=False):     # hypercube/src/air/builder.rs:L200-L250 (no trap code: mprotect off)
    (b.receive if receive else b.send)(SYSCALL, [clk_high, clk_low, syscall_id] + list(arg1) + list(arg2), mult)


# ---------------------------------------------------------------------------------------------------------------------
def syscall_chip(kind):                                                                   # syscall/chip.rs:L299-L420
    name = "SyscallCore" if kind == "core" else "SyscallPrecompile"
    b, c, _ = _chip(name, 10)
    L = S(("clk_high", 1), ("clk_low", 1), ("syscall_id", 1), ("arg1", 3), ("arg2", 3), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    cube = L.is_real * L.is_real * L.is_real
    b.assert_eq(cube, cube)                                                               # the reference's degree-3 filler (L323-L326)
    slice_range_check_u8(b, [L.syscall_id, b.const(0)], L.is_real)                        # [syscall_id, trap_code = 0]
    slice_range_check_u16(b, [L.arg1[0]], L.is_real)
    send_syscall(b, L.clk_high, L.clk_low, L.syscall_id, L.arg1, L.arg2, L.is_real, receive=(kind == "core"))
    # the core shard SENDS the syscall to the global table, the precompile shard RECEIVES it there
    b.send(GLOBAL, [L.clk_high, L.clk_low, L.syscall_id + L.arg1[0] * (1 << 8), L.arg1[1], L.arg1[2], L.arg2[0], L.arg2[1], L.arg2[2],
                    1 if kind == "core" else 0, 0 if kind == "core" else 1, SYSCALL], L.is_real)
    return _done(b, c)


def memory_global_chip(kind):                                                             # memory/global.rs:L307-L474
    init = kind == "init"
    b, c, _ = _chip("MemoryGlobalInit" if init else "MemoryGlobalFinalize", 30)
    L = S(("clk_high", 1), ("clk_low", 1), ("index", 1), ("prev_addr", 3), ("addr", 3), ("lt_cols", LT_UNSIGNED), ("value", 4),
          ("value_lower", 1), ("value_upper", 1), ("is_real", 1), ("is_comp", 1), ("prev_valid", 1), ("is_prev_addr_zero", IS_ZERO),
          ("is_index_zero", IS_ZERO))(c)
    b.assert_bool(L.is_real)
    slice_range_check_u16(b, L.value, L.is_real)
    slice_range_check_u16(b, L.prev_addr, L.is_real)
    slice_range_check_u16(b, L.addr, L.is_real)
    b.assert_eq(L.value[2], L.value_lower + L.value_upper * (1 << 8))
    slice_range_check_u8(b, [L.value_lower, L.value_upper], L.is_real)
    control = MEMORY_GLOBAL_INIT_CONTROL if init else MEMORY_GLOBAL_FINALIZE_CONTROL
    b.receive(control, [L.index] + L.prev_addr + [L.prev_valid], L.is_real)
    b.send(control, [L.index + 1] + L.addr + [L.is_comp], L.is_real)
    limbs = [L.addr[0], L.addr[1], L.addr[2], L.value[0] + L.value_lower * (1 << 16), L.value[1] + L.value_upper * (1 << 16), L.value[3]]
    if init:
        b.send(GLOBAL, [0, 0] + limbs + [1, 0, MEMORY], L.is_real)
    else:
        b.send(GLOBAL, [L.clk_high, L.clk_low] + limbs + [0, 1, MEMORY], L.is_real)
    eval_is_zero(b, L.prev_addr[0] + L.prev_addr[1] + L.prev_addr[2], L.is_prev_addr_zero, L.is_real)
    eval_is_zero(b, L.index, L.is_index_zero, L.is_real)
    b.assert_eq(L.is_comp, L.is_real * (1 - L.is_prev_addr_zero.result * L.is_index_zero.result))
    b.assert_bool(L.is_comp)
    eval_lt_unsigned(b, L.prev_addr + [b.const(0)], L.addr + [b.const(0)], L.lt_cols, L.is_comp)
    b.when(L.is_comp).assert_one(L.lt_cols.bit)
    is_not_comp = L.is_real - L.is_comp
    b.when(is_not_comp).assert_zero(L.addr[0] + L.addr[1] + L.addr[2])
    b.when(is_not_comp).assert_word_zero(L.value)
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
def eval_field_word_range_check(b, value, bit, is_real):                                 # operations/sp1_field_word.rs:L48-L88
    TOP_LIMB = (P - 1) >> 16
    b.assert_bool(is_real)
    b.when(is_real).assert_zero(value[2])
    b.when(is_real).assert_zero(value[3])
    eval_compare_u16(b, value[1], b.const(TOP_LIMB), bit, is_real)
    b.when(is_real).when_not(bit).assert_eq(value[1], TOP_LIMB)
    b.when(is_real).when_not(bit).assert_zero(value[0])


def word_reduce(b, w):                                                                    # hypercube/src/word.rs:L77-L80
    return w[0] + w[1] * (1 << 16) + w[2] * ((1 << 32) % P) + w[3] * ((1 << 48) % P)


def syscall_instrs_chip():                                                                # syscall/instructions/air.rs:L28-L595
    b, c, _ = _chip("SyscallInstrs", 65)
    L = S(("state", CPU_STATE), ("adapter", R_TYPE), ("next_pc", 3), ("is_halt", 1), ("op_a_value", 4), ("a_low_bytes", U16_TO_U8),
          ("is_enter_unconstrained", IS_ZERO), ("is_hint_len", IS_ZERO), ("is_halt_check", IS_ZERO), ("is_commit", IS_ZERO),
          ("is_commit_deferred_proofs", IS_ZERO), ("index_bitmap", 8), ("expected_public_values_digest", 4), ("op_b_range_check", 1),
          ("op_c_range_check", 1), ("is_real", 1))(c)
    ad = L.adapter
    prev_a, op_b, op_c = ad.op_a_memory.prev_value, ad.op_b_memory.prev_value, ad.op_c_memory.prev_value
    a = u16_to_u8_safe(b, prev_a, L.a_low_bytes.low_bytes, L.is_real)
    b.assert_bool(L.is_real)
    syscall_id, send_to_table = a[0], a[1]
    # eval_is_halt_syscall (L516-L543)
    eval_is_zero(b, syscall_id - SYS_HALT, L.is_halt_check, L.is_real)
    b.assert_eq(L.is_halt, L.is_halt_check.result * L.is_real)
    eval_cpu_state(b, L.state, L.next_pc, CLK_INC + 256, L.is_real)
    eval_r_type(b, L.state, OPC["ECALL"], L.op_a_value, ad, L.is_real, L.is_real)
    b.when(L.is_real).assert_zero(ad.op_a_0)
    jump = L.is_halt
    for i, want in enumerate(next_pc_inc(L.state)):
        b.when(L.is_real).when(1 - jump).assert_eq(L.next_pc[i], want)
    # eval_ecall (L160-L268)
    b.when_not(L.is_real).assert_zero(send_to_table)
    b.when_not(L.is_real).assert_zero(L.is_halt)
    b.when_not(L.is_real).assert_zero(L.is_commit_deferred_proofs.result)
    b.when(send_to_table).assert_zero(op_b[3])
    b.when(send_to_table).assert_zero(op_c[3])
    b.assert_bool(send_to_table)
    send_syscall(b, L.state.clk_high, clk_low_of(L.state), syscall_id, op_b[:3], op_c[:3], send_to_table)
    eval_field_word_range_check(b, op_b, L.op_b_range_check, L.is_halt)
    eval_field_word_range_check(b, op_c, L.op_c_range_check, L.is_commit_deferred_proofs.result)
    eval_is_zero(b, syscall_id - SYS_ENTER_UNCONSTRAINED, L.is_enter_unconstrained, L.is_real)
    eval_is_zero(b, syscall_id - SYS_HINT_LEN, L.is_hint_len, L.is_real)
    b.when(L.is_real).when(L.is_enter_unconstrained.result).assert_word_eq(L.op_a_value, [0, 0, 0, 0])
    b.when(L.is_real).when_not(L.is_enter_unconstrained.result + L.is_hint_len.result).assert_word_eq(L.op_a_value, prev_a)
    slice_range_check_u16(b, L.op_a_value, L.is_real)
    # eval_commit (L271-L374); get_is_commit_related_syscall (L547-L594)
    commit_digest = [[b.public(PV_COMMITTED_VALUE_DIGEST + 4 * w + i) for i in range(4)] for w in range(8)]
    deferred_digest = [b.public(PV_DEFERRED_PROOFS_DIGEST + i) for i in range(8)]
    eval_is_zero(b, syscall_id - SYS_COMMIT, L.is_commit, L.is_real)
    eval_is_zero(b, syscall_id - SYS_COMMIT_DEFERRED_PROOFS, L.is_commit_deferred_proofs, L.is_real)
    is_commit, is_cdp = L.is_commit.result, L.is_commit_deferred_proofs.result
    b.when(is_commit).assert_one(b.public(PV_COMMIT_SYSCALL))
    b.when(is_cdp).assert_one(b.public(PV_COMMIT_DEFERRED_SYSCALL))
    bitmap_sum = b.const(0)
    for bit in L.index_bitmap:
        b.when(L.is_real).assert_bool(bit)
        bitmap_sum = bitmap_sum + bit
    b.when(L.is_real).when(is_commit + is_cdp).assert_one(bitmap_sum)
    b.when(L.is_real).when(1 - (is_commit + is_cdp)).assert_zero(bitmap_sum)
    for i, bit in enumerate(L.index_bitmap):
        b.when(L.is_real).when(bit).assert_eq(op_b[0], i)
    b.when(L.is_real).when(is_commit + is_cdp).assert_zero(op_b[1] + op_b[2] + op_b[3])
    index_array = lambda arr: sum((v * bit for v, bit in zip(arr[1:], L.index_bitmap[1:])), arr[0] * L.index_bitmap[0])   # builder.rs:L96-L108
    expected = [index_array([w[i] for w in commit_digest]) for i in range(4)]
    expected_word = [expected[0] + expected[1] * (1 << 8), expected[2] + expected[3] * (1 << 8), b.const(0), b.const(0)]
    b.assert_bool(is_commit)
    for i in range(4):
        b.when(is_commit).assert_eq(expected[i], L.expected_public_values_digest[i])
    slice_range_check_u8(b, L.expected_public_values_digest, is_commit)
    b.when(L.is_real).when(is_commit).assert_word_eq(expected_word, op_c)
    b.when(L.is_real).when(is_cdp).assert_eq(index_array(deferred_digest), word_reduce(b, op_c))
    # eval_halt_unimpl (L471-L492)
    b.when(L.is_halt).assert_eq(L.next_pc[0], HALT_PC)
    b.when(L.is_halt).assert_zero(L.next_pc[1])
    b.when(L.is_halt).assert_zero(L.next_pc[2])
    b.when(L.is_halt).assert_eq(word_reduce(b, op_b), b.public(PV_EXIT_CODE))
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
def alu_x0_chip():                                                                        # alu/alu_x0.rs:L236-L340
    """Every ALU instruction whose destination is x0: the result is discarded, the row only ties the instruction to the program
    table and performs the register accesses (op_a is 'written' with its previous value, which `op_a_0` forces to zero)."""
    b, c, _ = _chip("AluX0", 34)
    L = S(("state", CPU_STATE), ("adapter", ALU_TYPE), ("opcode", 1), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    b.when(L.is_real).assert_one(L.adapter.op_a_0)
    b.when_not(L.is_real).assert_zero(L.adapter.op_a_0)
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    send_byte(b, B_LTU, 1, L.opcode, 29, L.is_real)
    eval_alu_type(b, L.state, L.opcode, L.adapter.op_a_memory.prev_value, L.adapter, L.is_real, L.is_real)
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
def divrem_chip():                                                                        # alu/divrem/mod.rs:L597-L1313
    b, c, _ = _chip("DivRem", 246)
    L = S(("state", CPU_STATE), ("adapter", R_TYPE), ("a", 4), ("b", 4), ("c", 4), ("quotient", 4), ("quotient_comp", 4),
          ("remainder_comp", 4), ("remainder", 4), ("abs_remainder", 4), ("abs_c", 4), ("max_abs_c_or_1", 4), ("c_times_quotient", 8),
          ("c_times_quotient_lower", MUL_OP), ("c_times_quotient_upper", MUL_OP), ("c_neg_operation", ADD_OP),
          ("rem_neg_operation", ADD_OP), ("remainder_lt_operation", LT_UNSIGNED), ("carry", 8), ("is_c_0", IS_ZERO_WORD),
          ("is_div", 1), ("is_divu", 1), ("is_rem", 1), ("is_remu", 1), ("is_divw", 1), ("is_remw", 1), ("is_divuw", 1), ("is_remuw", 1),
          ("is_overflow", 1), ("is_overflow_b", IS_ZERO_WORD), ("is_overflow_c", IS_ZERO_WORD), ("b_msb", 1), ("rem_msb", 1), ("c_msb", 1),
          ("quot_msb", 1), ("b_neg", 1), ("b_neg_not_overflow", 1), ("b_not_neg_not_overflow", 1), ("is_real_not_word", 1), ("rem_neg", 1),
          ("c_neg", 1), ("abs_c_alu_event", 1), ("abs_rem_alu_event", 1), ("is_real", 1), ("remainder_check_multiplicity", 1))(c)
    ad = L.adapter
    op_b, op_c = ad.op_b_memory.prev_value, ad.op_c_memory.prev_value
    is_word = L.is_divw + L.is_remw + L.is_divuw + L.is_remuw
    is_not_word = L.is_divu + L.is_remu + L.is_div + L.is_rem
    is_signed_word = L.is_divw + L.is_remw
    is_unsigned_word = L.is_divuw + L.is_remuw
    is_signed_type = L.is_div + L.is_rem + L.is_divw + L.is_remw
    b.assert_eq(L.is_real_not_word, L.is_real * (1 - is_word))
    for msb, neg in ((L.b_msb, L.b_neg), (L.rem_msb, L.rem_neg), (L.c_msb, L.c_neg)):
        b.assert_eq(msb * is_signed_type, neg)
    for i in range(2):
        b.assert_eq(op_b[i], L.b[i])
        b.assert_eq(op_c[i], L.c[i])
    for i in range(2, 4):
        b.assert_eq(L.b[i], op_b[i] * (1 - is_word) + L.b_neg * is_word * U16_MAX)
        b.assert_eq(L.c[i], op_c[i] * (1 - is_word) + L.c_neg * is_word * U16_MAX)
    for comp, full, msb in ((L.quotient_comp, L.quotient, L.quot_msb), (L.remainder_comp, L.remainder, L.rem_msb)):
        for i in range(2):
            b.assert_eq(comp[i], full[i])
        for i in range(2, 4):
            b.when(is_unsigned_word).assert_eq(comp[i], 0)
            b.when(is_signed_word).assert_eq(comp[i], msb * U16_MAX)
            b.when(is_word).assert_eq(full[i], msb * U16_MAX)
            b.when(is_not_word).assert_eq(comp[i], full[i])
    # c * quotient through two MulOperations (L702-L754)
    zero = b.const(0)
    eval_mul(b, L.c_times_quotient[:4], L.quotient_comp, L.c, L.c_times_quotient_lower, L.is_real, L.is_real, zero, zero, zero, zero)
    is_mulh, is_mulhu = L.is_div + L.is_rem, L.is_divu + L.is_remu
    eval_mul(b, L.c_times_quotient[4:], L.quotient_comp, L.c, L.c_times_quotient_upper, L.is_real_not_word, zero, is_mulh, zero, is_mulhu, zero)
    # overflow (L756-L838)
    eval_is_equal_word(b, op_b, word_of_u64(1 << 63), L.is_overflow_b, L.is_real_not_word)
    eval_is_equal_word(b, op_c, word_of_u64((1 << 64) - 1), L.is_overflow_c, L.is_real_not_word)
    tb, tc = [op_b[0], op_b[1], zero, zero], [op_c[0], op_c[1], zero, zero]
    eval_is_equal_word(b, tb, word_of_u64(1 << 31), L.is_overflow_b, is_word)
    eval_is_equal_word(b, tc, word_of_u64((1 << 32) - 1), L.is_overflow_c, is_word)
    b.assert_eq(L.is_overflow, L.is_overflow_b.result * L.is_overflow_c.result * is_signed_type)
    b.assert_eq(L.b_neg_not_overflow, L.b_neg * (1 - L.is_overflow))
    b.assert_eq(L.b_not_neg_not_overflow, (1 - L.b_neg) * (1 - L.is_overflow))
    for i in range(4):
        b.when(L.is_overflow).assert_eq(L.quotient[i], L.b[i])
        b.when(L.is_overflow).assert_eq(L.remainder[i], 0)
    # c * quotient + remainder == b (L840-L886)
    sign_extension = L.rem_neg * U16_MAX
    acc = []
    for i in range(8):
        v = L.c_times_quotient[i] + (L.remainder_comp[i] if i < 4 else sign_extension)
        v = v - L.carry[i] * (1 << 16)
        if i > 0:
            v = v + L.carry[i - 1]
        acc.append(v)
    for i in range(8):
        b.when_not(L.is_overflow).assert_eq(L.b[i] if i < 4 else L.b_neg * U16_MAX, acc[i])
    slice_range_check_u16(b, acc, L.is_real)
    for i in range(4):
        b.when(L.is_divu + L.is_div + L.is_divw + L.is_divuw).assert_eq(L.quotient[i], L.a[i])
        b.when(L.is_remu + L.is_rem + L.is_remw + L.is_remuw).assert_eq(L.remainder[i], L.a[i])
    rem_limb_sum = L.remainder[0] + L.remainder[1] + L.remainder[2] + L.remainder[3]
    b.when(L.rem_neg).assert_one(L.b_neg)
    b.when(rem_limb_sum).when(1 - L.rem_neg).assert_zero(L.b_neg)
    # division by zero (L922-L945)
    eval_is_zero_word(b, L.c, L.is_c_0, L.is_real)
    for i in range(4):
        b.when(L.is_c_0.result).assert_eq(L.quotient[i], U16_MAX)
    for i in range(4):
        b.when(L.is_c_0.result).assert_eq(L.remainder_comp[i], L.b[i])
    # |remainder| < |c| (L947-L1036)
    for i in range(4):
        b.when_not(L.c_neg).assert_eq(L.c[i], L.abs_c[i])
        b.when_not(L.rem_neg).assert_eq(L.remainder_comp[i], L.abs_remainder[i])
    eval_add(b, L.c, L.abs_c, L.c_neg_operation.value, L.abs_c_alu_event)
    slice_range_check_u16(b, L.abs_c, L.is_real)
    b.when(L.abs_c_alu_event).assert_word_eq([0, 0, 0, 0], L.c_neg_operation.value)
    eval_add(b, L.remainder_comp, L.abs_remainder, L.rem_neg_operation.value, L.abs_rem_alu_event)
    slice_range_check_u16(b, L.abs_remainder, L.is_real)
    b.when(L.abs_rem_alu_event).assert_word_eq([0, 0, 0, 0], L.rem_neg_operation.value)
    b.assert_eq(L.abs_c_alu_event, L.c_neg * L.is_real)
    b.assert_eq(L.abs_rem_alu_event, L.rem_neg * L.is_real)
    is0 = L.is_c_0.result
    want = [is0 * 1 + (1 - is0) * L.abs_c[0]] + [(1 - is0) * L.abs_c[i] for i in range(1, 4)]
    for i in range(4):
        b.assert_eq(L.max_abs_c_or_1[i], want[i])
    b.assert_eq((1 - is0) * L.is_real, L.remainder_check_multiplicity)
    eval_lt_unsigned(b, L.abs_remainder, L.max_abs_c_or_1, L.remainder_lt_operation, L.remainder_check_multiplicity)
    b.when(L.remainder_check_multiplicity).assert_eq(1, L.remainder_lt_operation.bit)
    # MSBs (L1038-L1097)
    eval_msb(b, op_b[3], L.b_msb, L.is_real_not_word)
    eval_msb(b, op_c[3], L.c_msb, L.is_real_not_word)
    eval_msb(b, L.remainder[3], L.rem_msb, L.is_real_not_word)
    eval_msb(b, op_b[1], L.b_msb, is_word)
    eval_msb(b, op_c[1], L.c_msb, is_word)
    eval_msb(b, L.remainder[1], L.rem_msb, is_word)
    eval_msb(b, L.quotient[1], L.quot_msb, is_word)
    slice_range_check_u16(b, L.quotient, L.is_real)
    slice_range_check_u16(b, L.remainder, L.is_real)
    for carry in L.carry:
        b.assert_bool(carry)
    slice_range_check_u16(b, L.c_times_quotient, L.is_real)
    for flag in (L.is_div, L.is_divu, L.is_rem, L.is_remu, L.is_divw, L.is_remw, L.is_divuw, L.is_remuw, L.is_overflow, L.is_real_not_word,
                 L.b_neg, L.b_neg_not_overflow, L.b_not_neg_not_overflow, L.rem_neg, L.c_neg, L.is_real, L.abs_c_alu_event,
                 L.abs_rem_alu_event):
        b.assert_bool(flag)
    b.assert_eq(1, L.is_divu + L.is_remu + L.is_div + L.is_rem + L.is_divw + L.is_remw + L.is_divuw + L.is_remuw)
    opcode = (L.is_divu * OPC["DIVU"] + L.is_remu * OPC["REMU"] + L.is_div * OPC["DIV"] + L.is_rem * OPC["REM"] + L.is_divw * OPC["DIVW"] +
              L.is_remw * OPC["REMW"] + L.is_divuw * OPC["DIVUW"] + L.is_remuw * OPC["REMUW"])
    eval_cpu_state(b, L.state, next_pc_inc(L.state), CLK_INC, L.is_real)
    b.assert_zero(ad.op_a_0)
    eval_r_type(b, L.state, opcode, L.a, ad, L.is_real, L.is_real)
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
KECCAK_ROUNDS, U64_LIMBS = 24, 4
# rotation offsets r[x][y] of Keccak-f[1600] (the `R` table of p3-keccak-air; FIPS 202 §3.2.2)
KECCAK_R = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
             0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
             0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
             0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]

_grid = lambda n: (lambda c, p: [[c.arr(n, p + "%d.%d" % (y, x)) for x in range(5)] for y in range(5)])     # [y][x][n]
KECCAK_COLS = S(("step_flags", 24), ("export", 1), ("preimage", _grid(4)), ("a", _grid(4)),
                ("c", lambda c, p: [c.arr(64, p + "%d" % x) for x in range(5)]), ("c_prime", lambda c, p: [c.arr(64, p + "%d" % x) for x in range(5)]),
                ("a_prime", _grid(64)), ("a_prime_prime", _grid(4)), ("a_prime_prime_0_0_bits", 64), ("a_prime_prime_prime_0_0_limbs", 4))


def keccak_b(k, x, y, z):
    """`KeccakCols::b`: B[x, y] is a rotation of A'[(x + 3 y) % 5, x] (B[y, 2x + 3y] = ROT(A'[x, y], r[x, y]))."""
    xa, ya = (x + 3 * y) % 5, x
    return k.a_prime[ya][xa][(z + 64 - KECCAK_R[xa][ya]) % 64]


def keccak_permute_chip():                                                                # keccak256/air.rs:L29-L200
    b, c, _ = _chip("KeccakPermute", 2640)
    L = S(("keccak", KECCAK_COLS), ("clk_high", 1), ("clk_low", 1), ("state_addr", 3), ("index", 1), ("is_real", 1))(c)
    k = L.keccak
    b.assert_bool(L.is_real)
    b.air.hint_keccak(0)                                                                  # a prover may evaluate what follows with fused pieces (sp1_amd/csrc/zc_keccak.hpp)
    andn = lambda x, y: y - x * y
    xor = lambda x, y: x + y - x * (y * 2)
    xor3 = lambda x, y, z: xor(x, xor(y, z))
    sum_flags, computed_index = b.const(0), b.const(0)
    for i in range(KECCAK_ROUNDS):
        b.assert_bool(k.step_flags[i])
        sum_flags = sum_flags + k.step_flags[i]
        computed_index = computed_index + k.step_flags[i] * i
    b.assert_one(sum_flags)
    b.when(L.is_real).assert_eq(computed_index, L.index)
    for x in range(5):
        for z in range(64):
            b.assert_bool(k.c[x][z])
            b.assert_eq(k.c_prime[x][z], xor3(k.c[x][z], k.c[(x + 4) % 5][z], k.c[(x + 1) % 5][(z + 63) % 64]))
    for y in range(5):
        for x in range(5):
            for limb in range(U64_LIMBS):
                acc = b.const(0)
                for z in reversed(range(limb * 16, (limb + 1) * 16)):
                    b.assert_bool(k.a_prime[y][x][z])
                    acc = acc * 2 + xor3(k.a_prime[y][x][z], k.c[x][z], k.c_prime[x][z])
                b.assert_eq(acc, k.a[y][x][limb])
    for x in range(5):
        for z in range(64):
            diff = sum((k.a_prime[y][x][z] for y in range(1, 5)), k.a_prime[0][x][z]) - k.c_prime[x][z]
            b.assert_zero(diff * (diff - 2) * (diff - 4))
    for y in range(5):
        for x in range(5):
            for limb in range(U64_LIMBS):
                acc = b.const(0)
                for z in reversed(range(limb * 16, (limb + 1) * 16)):
                    acc = acc * 2 + xor(keccak_b(k, x, y, z), andn(keccak_b(k, (x + 1) % 5, y, z), keccak_b(k, (x + 2) % 5, y, z)))
                b.assert_eq(acc, k.a_prime_prime[y][x][limb])
    for limb in range(U64_LIMBS):
        acc = b.const(0)
        for z in reversed(range(limb * 16, (limb + 1) * 16)):
            b.assert_bool(k.a_prime_prime_0_0_bits[z])
            acc = acc * 2 + k.a_prime_prime_0_0_bits[z]
        b.assert_eq(acc, k.a_prime_prime[0][0][limb])

    def xored_bit(i):
        rc_bit = b.const(0)
        for r in range(KECCAK_ROUNDS):
            rc_bit = rc_bit + k.step_flags[r] * ((KECCAK_RC[r] >> i) & 1)
        return xor(k.a_prime_prime_0_0_bits[i], rc_bit)
    for limb in range(U64_LIMBS):
        acc = b.const(0)
        for z in reversed(range(limb * 16, (limb + 1) * 16)):
            acc = acc * 2 + xored_bit(z)
        b.assert_eq(acc, k.a_prime_prime_prime_0_0_limbs[limb])
    head = [L.clk_high, L.clk_low] + L.state_addr
    b.receive(KECCAK, head + [L.index] + [k.a[y][x][l] for y in range(5) for x in range(5) for l in range(4)], L.is_real)
    appp = lambda y, x, l: k.a_prime_prime_prime_0_0_limbs[l] if (y, x) == (0, 0) else k.a_prime_prime[y][x][l]
    b.send(KECCAK, head + [L.index + 1] + [appp(y, x, l) for y in range(5) for x in range(5) for l in range(4)], L.is_real)
    return _done(b, c)


def eval_syscall_addr(b, length, cols, is_real):                                          # operations/syscall_addr.rs:L51-L93
    assert length % 8 == 0
    b.assert_bool(is_real)
    top = cols.addr[1] + cols.addr[2]
    b.assert_eq(cols.top_two_limb_min * top, is_real)
    eval_is_zero(b, top - 2 * U16_MAX, cols.top_two_limb_max, is_real)
    send_byte(b, B_RANGE, (cols.addr[0] + cols.top_two_limb_max.result * length) * INV(8), 13, 0, is_real)
    return cols.addr


def keccak_control_chip():                                                                # keccak256/controller.rs:L243-L398
    b, c, _ = _chip("KeccakPermuteControl", 634)
    L = S(("clk_high", 1), ("clk_low", 1), ("state_addr", SYSCALL_ADDR), ("addrs", lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(25)]),
          ("is_real", 1), ("initial_memory_access", lambda c_, p: [MEM_ACCESS(c_, p + "%d." % i) for i in range(25)]),
          ("final_memory_access", lambda c_, p: [MEM_ACCESS(c_, p + "%d." % i) for i in range(25)]),
          ("final_value", lambda c_, p: [c_.arr(4, p + "%d" % i) for i in range(25)]))(c)
    b.assert_bool(L.is_real)
    state_addr = eval_syscall_addr(b, 200, L.state_addr, L.is_real)
    is_not_trap = L.is_real
    send_syscall(b, L.clk_high, L.clk_low, SYS_KECCAK_PERMUTE, state_addr, [0, 0, 0], L.is_real, receive=True)
    head = [L.clk_high, L.clk_low] + list(state_addr)
    b.send(KECCAK, head + [0] + [v for acc in L.initial_memory_access for v in acc.prev_value], is_not_trap)
    b.receive(KECCAK, head + [24] + [v for w in L.final_value for v in w], is_not_trap)
    for i in range(25):
        eval_addr_add(b, list(state_addr) + [b.const(0)], word_of_u64(8 * i), L.addrs[i].value, L.is_real)
    for i in range(25):
        eval_memory_access(b, L.clk_high, L.clk_low, L.addrs[i].value, L.initial_memory_access[i], L.initial_memory_access[i].prev_value,
                           is_not_trap)
        eval_memory_access(b, L.clk_high, L.clk_low + 1, L.addrs[i].value, L.final_memory_access[i], L.final_value[i], is_not_trap)
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
# SHA-256 precompiles: operations on u32 values held as two 16-bit limbs
FIXED_ROT = S(("value", 2), ("higher_limb", 2))                                          # operations/fixed_rotate_right.rs:L14-L20, fixed_shift_right.rs
U32_TO_U8 = S(("low_bytes", 2),)                                                         # operations/u32_operation.rs
BYTE_OP_U32 = S(("b_low_bytes", U32_TO_U8), ("c_low_bytes", U32_TO_U8), ("value", 4))    # xor_u32.rs / and_u32.rs
HALF_WORD = S(("value", 2),)                                                             # not_u32.rs, add4.rs, add5.rs, add_u32.rs
CLK_OP = S(("next_clk_16_24", 1), ("next_clk_0_16", 1), ("is_overflow", 1))              # operations/clk.rs
SHA_EXTEND, SHA_COMPRESS = 10, 11                                                        # hypercube/src/lookup/interaction.rs:L45-L49
SYS_SHA_EXTEND, SYS_SHA_COMPRESS = 0x05, 0x06
SHA_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]   # FIPS 180-4 §4.2.2


def eval_fixed_rotate_right(b, x, rotation, cols, is_real, shift=False):                   # fixed_rotate_right.rs:L63-L112 / fixed_shift_right.rs:L66-L115
    b.assert_bool(is_real)
    nl, nb = rotation // 16, rotation % 16
    mult = 1 << (16 - nb)
    limbs = [x[i + nl] if i + nl < 2 else b.const(0) for i in range(2)] if shift else [x[nl % 2], x[(1 + nl) % 2]]
    lower = []
    for i in range(2):
        lower.append(limbs[i] - cols.higher_limb[i] * (1 << nb))
        send_byte(b, B_RANGE, lower[i], nb, 0, is_real)
        send_byte(b, B_RANGE, cols.higher_limb[i], 16 - nb, 0, is_real)
    if shift:
        b.when(is_real).assert_eq(cols.value[1], cols.higher_limb[1])
    else:
        b.when(is_real).assert_eq(cols.value[1], cols.higher_limb[1] + lower[0] * mult)
    b.when(is_real).assert_eq(cols.value[0], cols.higher_limb[0] + lower[1] * mult)


def _u32_to_u8(x, cols):                                                                  # u32_operation.rs: eval_u32_to_u8_unsafe
    out = []
    for i in range(2):
        out += [cols.low_bytes[i], (x[i] - cols.low_bytes[i]) * INV(1 << 8)]
    return out


def eval_byte_op_u32(b, opcode, x, y, cols, is_real):                                     # xor_u32.rs:L61-L92 / and_u32.rs:L58-L89
    b.assert_bool(is_real)
    xb, yb = _u32_to_u8(x, cols.b_low_bytes), _u32_to_u8(y, cols.c_low_bytes)
    for i in range(4):
        send_byte(b, opcode, cols.value[i], xb[i], yb[i], is_real)
    return [cols.value[0] + cols.value[1] * (1 << 8), cols.value[2] + cols.value[3] * (1 << 8)]


def eval_not_u32(b, a, cols, is_real):                                                    # not_u32.rs:L27-L39
    for i in range(2):
        b.when(is_real).assert_eq(cols.value[i] + a[i], U16_MAX)


def eval_add_n(b, words, cols, is_real):                                                  # add4.rs:L59-L88, add5.rs:L66-L96
    b.assert_bool(is_real)
    carry, carries = b.const(0), []
    for i in range(2):
        acc = carry - cols.value[i]
        for w in words:
            acc = acc + w[i]
        carry = acc * INV(1 << 16)
        carries.append(carry)
    slice_range_check_u16(b, cols.value, is_real)
    slice_range_check_u8(b, carries, is_real)


def eval_add_u32(b, x, y, cols, is_real):                                                 # add_u32.rs:L36-L58
    b.assert_bool(is_real)
    w, carry = b.when(is_real), b.const(0)
    for i in range(2):
        carry = (x[i] + y[i] - cols.value[i] + carry) * INV(1 << 16)
        w.assert_bool(carry)
    slice_range_check_u16(b, cols.value, is_real)


def eval_clk(b, clk_low, increment, cols, is_real):                                       # clk.rs:L61-L88; returns (is_overflow, next clk_low)
    b.assert_bool(is_real)
    b.assert_bool(cols.is_overflow)
    nxt = cols.next_clk_0_16 + cols.next_clk_16_24 * (1 << 16)
    b.when(is_real).assert_eq(clk_low + increment - cols.is_overflow * (1 << 24), nxt)
    send_byte(b, B_RANGE, cols.next_clk_0_16, 16, 0, is_real)
    slice_range_check_u8(b, [cols.next_clk_16_24, b.const(0)], is_real)
    return cols.is_overflow, nxt


def _expr_word(b, e):                                                                     # Word::extend_expr: (e, 0, 0, 0)
    return [e, b.const(0), b.const(0), b.const(0)]


def sha_extend_chip():                                                                    # sha256/extend/air.rs:L28-L317
    b, c, _ = _chip("ShaExtend", 128)
    L = S(("clk_high", 1), ("clk_low", 1), ("next_clk", CLK_OP), ("w_ptr", 3), ("w_i_minus_15_ptr", ADDR_ADD_OP), ("w_i_minus_2_ptr", ADDR_ADD_OP),
          ("w_i_minus_16_ptr", ADDR_ADD_OP), ("w_i_minus_7_ptr", ADDR_ADD_OP), ("w_i_ptr", ADDR_ADD_OP), ("i", 1),
          ("w_i_minus_15", MEM_ACCESS), ("w_i_minus_15_rr_7", FIXED_ROT), ("w_i_minus_15_rr_18", FIXED_ROT), ("w_i_minus_15_rs_3", FIXED_ROT),
          ("s0_intermediate", BYTE_OP_U32), ("s0", BYTE_OP_U32),
          ("w_i_minus_2", MEM_ACCESS), ("w_i_minus_2_rr_17", FIXED_ROT), ("w_i_minus_2_rr_19", FIXED_ROT), ("w_i_minus_2_rs_10", FIXED_ROT),
          ("s1_intermediate", BYTE_OP_U32), ("s1", BYTE_OP_U32), ("w_i_minus_16", MEM_ACCESS), ("w_i_minus_7", MEM_ACCESS), ("s2", HALF_WORD),
          ("w_i", MEM_ACCESS), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    head = [L.clk_high, L.clk_low] + L.w_ptr
    b.receive(SHA_EXTEND, head + [L.i], L.is_real)
    b.send(SHA_EXTEND, head + [L.i + 1], L.is_real)
    send_byte(b, B_LTU, 1, L.i - 16, 48, L.is_real)
    overflow, next_low = eval_clk(b, L.clk_low, L.i - 16, L.next_clk, L.is_real)
    next_high = L.clk_high + overflow
    ptr = L.w_ptr + [b.const(0)]
    halves = {}
    for name, off in (("w_i_minus_15", 15), ("w_i_minus_2", 2), ("w_i_minus_16", 16), ("w_i_minus_7", 7)):
        addr, acc = getattr(L, name + "_ptr"), getattr(L, name)
        eval_addr_add(b, ptr, _expr_word(b, (L.i - off) * 8), addr.value, L.is_real)
        eval_memory_access(b, next_high, next_low, addr.value, acc, acc.prev_value, L.is_real)      # a read
        halves[name] = [acc.prev_value[0], acc.prev_value[1]]
        b.assert_zero(acc.prev_value[2])
        b.assert_zero(acc.prev_value[3])
    eval_fixed_rotate_right(b, halves["w_i_minus_15"], 7, L.w_i_minus_15_rr_7, L.is_real)
    eval_fixed_rotate_right(b, halves["w_i_minus_15"], 18, L.w_i_minus_15_rr_18, L.is_real)
    eval_fixed_rotate_right(b, halves["w_i_minus_15"], 3, L.w_i_minus_15_rs_3, L.is_real, shift=True)
    t = eval_byte_op_u32(b, B_XOR, L.w_i_minus_15_rr_7.value, L.w_i_minus_15_rr_18.value, L.s0_intermediate, L.is_real)
    s0 = eval_byte_op_u32(b, B_XOR, t, L.w_i_minus_15_rs_3.value, L.s0, L.is_real)
    eval_fixed_rotate_right(b, halves["w_i_minus_2"], 17, L.w_i_minus_2_rr_17, L.is_real)
    eval_fixed_rotate_right(b, halves["w_i_minus_2"], 19, L.w_i_minus_2_rr_19, L.is_real)
    eval_fixed_rotate_right(b, halves["w_i_minus_2"], 10, L.w_i_minus_2_rs_10, L.is_real, shift=True)
    t = eval_byte_op_u32(b, B_XOR, L.w_i_minus_2_rr_17.value, L.w_i_minus_2_rr_19.value, L.s1_intermediate, L.is_real)
    s1 = eval_byte_op_u32(b, B_XOR, t, L.w_i_minus_2_rs_10.value, L.s1, L.is_real)
    eval_add_n(b, [halves["w_i_minus_16"], s0, halves["w_i_minus_7"], s1], L.s2, L.is_real)
    eval_addr_add(b, ptr, _expr_word(b, L.i * 8), L.w_i_ptr.value, L.is_real)
    eval_memory_access(b, next_high, next_low, L.w_i_ptr.value, L.w_i, [L.s2.value[0], L.s2.value[1], b.const(0), b.const(0)], L.is_real)
    return _done(b, c)


def sha_extend_control_chip():                                                            # sha256/extend/controller.rs:L172-L311
    b, c, _ = _chip("ShaExtendControl", 18)
    L = S(("clk_high", 1), ("clk_low", 1), ("w_ptr", SYSCALL_ADDR), ("w_16th_addr", ADDR_ADD_OP), ("w_17th_addr", ADDR_ADD_OP),
          ("w_64th_addr", ADDR_ADD_OP), ("is_real", 1))(c)
    b.assert_bool(L.is_real)
    w_ptr = eval_syscall_addr(b, 512, L.w_ptr, L.is_real)
    for cols, off in ((L.w_16th_addr, 15), (L.w_17th_addr, 16), (L.w_64th_addr, 63)):
        eval_addr_add(b, list(w_ptr) + [b.const(0)], word_of_u64(off * 8), cols.value, L.is_real)
    send_syscall(b, L.clk_high, L.clk_low, SYS_SHA_EXTEND, w_ptr, [0, 0, 0], L.is_real, receive=True)
    b.send(SHA_EXTEND, [L.clk_high, L.clk_low + 1] + list(w_ptr) + [16], L.is_real)
    b.receive(SHA_EXTEND, [L.clk_high, L.clk_low + 1] + list(w_ptr) + [64], L.is_real)
    return _done(b, c)


def sha_compress_chip():                                                                  # sha256/compress/air.rs:L34-L500
    b, c, _ = _chip("ShaCompress", 206)
    L = S(("clk_high", 1), ("clk_low", 1), ("w_ptr", 3), ("h_ptr", 3), ("index", 1), ("octet", 8), ("octet_num", 10), ("mem", MEM_ACCESS),
          ("mem_value", 2), ("mem_addr", 3), ("mem_addr_init", ADDR_ADD_OP), ("mem_addr_compress", ADDR_ADD_OP), ("mem_addr_finalize", ADDR_ADD_OP),
          ("a", 2), ("b", 2), ("c", 2), ("d", 2), ("e", 2), ("f", 2), ("g", 2), ("h", 2), ("k", 2),
          ("e_rr_6", FIXED_ROT), ("e_rr_11", FIXED_ROT), ("e_rr_25", FIXED_ROT), ("s1_intermediate", BYTE_OP_U32), ("s1", BYTE_OP_U32),
          ("e_and_f", BYTE_OP_U32), ("e_not", HALF_WORD), ("e_not_and_g", BYTE_OP_U32), ("ch", BYTE_OP_U32), ("temp1", HALF_WORD),
          ("a_rr_2", FIXED_ROT), ("a_rr_13", FIXED_ROT), ("a_rr_22", FIXED_ROT), ("s0_intermediate", BYTE_OP_U32), ("s0", BYTE_OP_U32),
          ("a_and_b", BYTE_OP_U32), ("a_and_c", BYTE_OP_U32), ("b_and_c", BYTE_OP_U32), ("maj_intermediate", BYTE_OP_U32), ("maj", BYTE_OP_U32),
          ("temp2", HALF_WORD), ("d_add_temp1", HALF_WORD), ("temp1_add_temp2", HALF_WORD), ("finalized_operand", 2), ("finalize_add", HALF_WORD),
          ("is_initialize", 1), ("is_compression", 1), ("is_finalize", 1), ("is_real", 1))(c)
    # eval_control_flow_flags (L52-L160)
    b.assert_bool(L.is_real)
    computed, s_ = b.const(0), b.const(0)
    for i in range(8):
        b.assert_bool(L.octet[i])
        s_ = s_ + L.octet[i]
        computed = computed + L.octet[i] * i
    b.assert_one(s_)
    s_ = b.const(0)
    for i in range(10):
        b.assert_bool(L.octet_num[i])
        s_ = s_ + L.octet_num[i]
        computed = computed + L.octet_num[i] * (8 * i)
    b.assert_one(s_)
    b.assert_eq(L.index, computed)
    b.assert_eq(L.is_initialize, L.octet_num[0] * L.is_real)
    mid = L.octet_num[1]
    for i in range(2, 9):
        mid = mid + L.octet_num[i]
    b.assert_eq(L.is_compression, mid * L.is_real)
    b.assert_eq(L.is_finalize, L.octet_num[9] * L.is_real)
    vars8 = [L.a, L.b, L.c, L.d, L.e, L.f, L.g, L.h]
    head = [L.clk_high, L.clk_low] + L.w_ptr + L.h_ptr
    flat = lambda ws: [x for w in ws for x in w]
    b.receive(SHA_COMPRESS, head + [L.index] + flat(vars8), L.is_real)
    b.send(SHA_COMPRESS, head + [L.index + 1] + flat(vars8), L.is_initialize + L.is_finalize)
    b.send(SHA_COMPRESS, head + [L.index + 1] + flat([L.temp1_add_temp2.value, L.a, L.b, L.c, L.d_add_temp1.value, L.e, L.f, L.g]), L.is_compression)
    # eval_memory (L162-L253)
    zero = b.const(0)
    mem_word = [L.mem_value[0], L.mem_value[1], zero, zero]
    eval_memory_access(b, L.clk_high, L.clk_low + L.is_compression + L.is_finalize * 2, L.mem_addr, L.mem, mem_word, L.is_real)
    b.when(L.is_initialize + L.is_compression).assert_word_eq(L.mem.prev_value, mem_word)
    b.assert_zero(L.mem.prev_value[2])
    b.assert_zero(L.mem.prev_value[3])
    b.when(L.is_initialize).assert_all_eq(L.mem_addr, L.mem_addr_init.value)
    b.when(L.is_compression).assert_all_eq(L.mem_addr, L.mem_addr_compress.value)
    b.when(L.is_finalize).assert_all_eq(L.mem_addr, L.mem_addr_finalize.value)
    eval_addr_add(b, L.h_ptr + [zero], _expr_word(b, L.index * 8), L.mem_addr_init.value, L.is_initialize)
    eval_addr_add(b, L.w_ptr + [zero], _expr_word(b, (L.index - 8) * 8), L.mem_addr_compress.value, L.is_compression)
    eval_addr_add(b, L.h_ptr + [zero], _expr_word(b, (L.index - 72) * 8), L.mem_addr_finalize.value, L.is_finalize)
    for i, v in enumerate(vars8):
        vw = [v[0], v[1], zero, zero]
        b.when(L.is_initialize * L.octet[i]).assert_word_eq(vw, L.mem.prev_value)
        b.when(L.is_initialize * L.octet[i]).assert_word_eq(vw, mem_word)
    b.when(L.is_finalize).assert_all_eq(L.mem_value, L.finalize_add.value)
    # eval_compression_ops (L255-L451)
    for i in range(64):
        b.when(L.octet_num[i // 8 + 1] * L.octet[i % 8]).assert_all_eq(L.k, [SHA_K[i] & 0xFFFF, SHA_K[i] >> 16])
    ic = L.is_compression
    eval_fixed_rotate_right(b, L.e, 6, L.e_rr_6, ic)
    eval_fixed_rotate_right(b, L.e, 11, L.e_rr_11, ic)
    eval_fixed_rotate_right(b, L.e, 25, L.e_rr_25, ic)
    t = eval_byte_op_u32(b, B_XOR, L.e_rr_6.value, L.e_rr_11.value, L.s1_intermediate, ic)
    s1 = eval_byte_op_u32(b, B_XOR, t, L.e_rr_25.value, L.s1, ic)
    e_and_f = eval_byte_op_u32(b, B_AND, L.e, L.f, L.e_and_f, ic)
    eval_not_u32(b, L.e, L.e_not, ic)
    e_not_and_g = eval_byte_op_u32(b, B_AND, L.e_not.value, L.g, L.e_not_and_g, ic)
    ch = eval_byte_op_u32(b, B_XOR, e_and_f, e_not_and_g, L.ch, ic)
    eval_add_n(b, [L.h, s1, ch, L.k, L.mem_value], L.temp1, ic)
    eval_fixed_rotate_right(b, L.a, 2, L.a_rr_2, ic)
    eval_fixed_rotate_right(b, L.a, 13, L.a_rr_13, ic)
    eval_fixed_rotate_right(b, L.a, 22, L.a_rr_22, ic)
    t = eval_byte_op_u32(b, B_XOR, L.a_rr_2.value, L.a_rr_13.value, L.s0_intermediate, ic)
    s0 = eval_byte_op_u32(b, B_XOR, t, L.a_rr_22.value, L.s0, ic)
    a_and_b = eval_byte_op_u32(b, B_AND, L.a, L.b, L.a_and_b, ic)
    a_and_c = eval_byte_op_u32(b, B_AND, L.a, L.c, L.a_and_c, ic)
    b_and_c = eval_byte_op_u32(b, B_AND, L.b, L.c, L.b_and_c, ic)
    t = eval_byte_op_u32(b, B_XOR, a_and_b, a_and_c, L.maj_intermediate, ic)
    maj = eval_byte_op_u32(b, B_XOR, t, b_and_c, L.maj, ic)
    eval_add_u32(b, s0, maj, L.temp2, ic)
    eval_add_u32(b, L.d, L.temp1.value, L.d_add_temp1, ic)
    eval_add_u32(b, L.temp1.value, L.temp2.value, L.temp1_add_temp2, ic)
    # eval_finalize_ops (L453-L482)
    filt = [b.const(0), b.const(0)]
    for flag, v in zip(L.octet, vars8):
        filt = [filt[0] + flag * v[0], filt[1] + flag * v[1]]
    b.when(L.is_finalize).assert_all_eq(filt, L.finalized_operand)
    eval_add_u32(b, [L.mem.prev_value[0], L.mem.prev_value[1]], L.finalized_operand, L.finalize_add, L.is_finalize)
    return _done(b, c)


def sha_compress_control_chip():                                                          # sha256/compress/controller.rs:L205-L345
    b, c, _ = _chip("ShaCompressControl", 53)
    L = S(("clk_high", 1), ("clk_low", 1), ("w_ptr", SYSCALL_ADDR), ("h_ptr", SYSCALL_ADDR), ("w_slice_end", ADDR_ADD_OP), ("h_slice_end", ADDR_ADD_OP),
          ("is_real", 1), ("initial_state", 16), ("final_state", 16))(c)
    b.assert_bool(L.is_real)
    w_ptr = eval_syscall_addr(b, 512, L.w_ptr, L.is_real)
    h_ptr = eval_syscall_addr(b, 64, L.h_ptr, L.is_real)
    eval_addr_add(b, list(w_ptr) + [b.const(0)], word_of_u64(63 * 8), L.w_slice_end.value, L.is_real)
    eval_addr_add(b, list(h_ptr) + [b.const(0)], word_of_u64(7 * 8), L.h_slice_end.value, L.is_real)
    send_syscall(b, L.clk_high, L.clk_low, SYS_SHA_COMPRESS, w_ptr, h_ptr, L.is_real, receive=True)
    head = [L.clk_high, L.clk_low] + list(w_ptr) + list(h_ptr)
    b.send(SHA_COMPRESS, head + [0] + L.initial_state, L.is_real)
    b.receive(SHA_COMPRESS, head + [80] + L.final_state, L.is_real)
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
# Field operations on byte limbs (operations/field/): polynomials over expressions, lowest coefficient first
def _poly_add(p, q):
    n = max(len(p), len(q))
    return [(p[i] if i < len(p) else 0) + (q[i] if i < len(q) else 0) for i in range(n)]


def _poly_sub(p, q):
    n = max(len(p), len(q))
    return [(p[i] if i < len(p) else 0) - (q[i] if i < len(q) else 0) for i in range(n)]


def _poly_mul(p, q):
    out = [0] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, c in enumerate(q):
            out[i + j] = out[i + j] + a * c
    return out


def _poly_scale(p, k):
    return [a * k for a in p]


WITNESS_OFFSET = 1 << 14                                                                  # curves/src/{uint256,weierstrass/*}.rs


def FIELD_OP(n_limbs, n_witness):                                                         # field_op.rs:L18-L24
    return S(("result", n_limbs), ("carry", n_limbs), ("witness", n_witness))


def FIELD_LT(n_limbs):                                                                    # field/range.rs:L18-L27
    return S(("byte_flags", n_limbs), ("lhs_comparison_byte", 1), ("rhs_comparison_byte", 1))


MEM_ACCESS_U8 = S(("memory_access", MEM_ACCESS), ("prev_value_u8", U16_TO_U8))            # memory/consistency/columns.rs:L38-L43


def eval_field_op_polynomials(b, cols, p_op, p_modulus, p_result, is_real, witness_offset=WITNESS_OFFSET, products=None):   # field_op.rs eval_with_polynomials + util_air.rs
    """The coefficients of p_op - p_result - carry * modulus - (witness - offset) * (x - 2^8) are zero. `products`: p_op is
    `p_op` + the sum over the terms given of the product of a term's two or three polynomials — the same polynomial, with its
    products named so that the constraint program can carry a hint for provers (AirProgram.hint_polynomial_identity) when every
    factor and everything else is affine in the row. p_modulus None: carry * modulus is among the terms (a modulus from memory)."""
    witness = [w - witness_offset for w in cols.witness]
    rhs = _poly_mul(witness, [-(1 << 8), 1])
    if products is None:
        p_vanishing = _poly_sub(_poly_sub(p_op, p_result), _poly_mul(cols.carry, p_modulus))
        for cst in _poly_sub(p_vanishing, rhs):
            b.assert_zero(cst)
    else:
        sym = lambda v: v if isinstance(v, Sym) else b.const(v)
        products = [tuple([sym(v) for v in f] for f in fs) for fs in products]
        products = [fs for fs in products if not any(all(v.is_const and v.const_value == 0 for v in f) for f in fs)]   # (a selector that is the constant 0)
        carry_mod = _poly_mul(cols.carry, p_modulus) if p_modulus is not None else []
        rest = _poly_sub(_poly_sub(_poly_sub(p_op, p_result), carry_mod), rhs)
        conv = []
        for fs in products:
            term = fs[0]
            for f in fs[1:]:
                term = _poly_mul(term, f)
            conv = _poly_add(conv, term)
        rest = [sym(v) for v in rest + [0] * (len(conv) - len(rest))]
        main_affine = lambda v: v.lin is not None and all(kind == "main" for kind, _ in v.lin[0])
        if isinstance(b, Builder) and len(products) <= 4 and all(main_affine(v) for fs in products for f in fs for v in f) and all(main_affine(v) for v in rest):
            b.air.hint_polynomial_identity([tuple([v.expr() for v in f] for f in fs) for fs in products], [v.expr() for v in rest])
        for k, v in enumerate(rest):
            b.assert_zero(conv[k] + v if k < len(conv) else v)
    slice_range_check_u8(b, cols.result, is_real)
    slice_range_check_u8(b, cols.carry, is_real)
    slice_range_check_u16(b, cols.witness, is_real)


def eval_field_lt(b, cols, lhs, rhs, is_real):                                            # field/range.rs:L64-L139
    s_ = b.const(0)
    for f in cols.byte_flags:
        b.when(is_real).assert_bool(f)
        s_ = s_ + f
    b.when(is_real).assert_one(s_)
    visited, lb, rb = b.const(0), b.const(0), b.const(0)
    for lbyte, rbyte, f in zip(reversed(lhs), reversed(rhs), reversed(cols.byte_flags)):
        visited = visited + f
        lb = lb + lbyte * f
        rb = rb + f * rbyte
        b.when(is_real).when_not(visited).assert_eq(lbyte, rbyte)
    b.when(is_real).assert_eq(cols.lhs_comparison_byte, lb)
    b.when(is_real).assert_eq(cols.rhs_comparison_byte, rb)
    send_byte(b, B_LTU, 1, cols.lhs_comparison_byte, cols.rhs_comparison_byte, is_real)


def generate_limbs(b, accesses, is_real):                                                 # air/mod.rs:L22-L43
    out = []
    for acc in accesses:
        out += u16_to_u8_safe(b, acc.memory_access.prev_value, acc.prev_value_u8.low_bytes, is_real)
    return out


def limbs_to_words(limbs):                                                                # utils/mod.rs:L26-L40
    return [[limbs[8 * w + 2 * k] + limbs[8 * w + 2 * k + 1] * (1 << 8) for k in range(4)] for w in range(len(limbs) // 8)]


SYS_UINT256_MUL = 0x1D


def uint256_mul_chip():                                                                   # syscall/precompiles/uint256/air.rs:L318-L517
    b, c, _ = _chip("Uint256MulMod", 371)
    accs = lambda n: (lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(n)])
    L = S(("clk_high", 1), ("clk_low", 1), ("x_ptr", SYSCALL_ADDR), ("y_ptr", SYSCALL_ADDR),
          ("x_addrs", lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(4)]),
          ("y_and_modulus_addrs", lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(8)]),
          ("x_memory", accs(4)), ("y_memory", accs(4)), ("modulus_memory", accs(4)), ("modulus_is_zero", IS_ZERO), ("modulus_is_not_zero", 1),
          ("output", FIELD_OP(32, 63)), ("output_range_check", FIELD_LT(32)), ("is_real", 1))(c)
    x_limbs = generate_limbs(b, L.x_memory, L.is_real)
    y_limbs = generate_limbs(b, L.y_memory, L.is_real)
    m_limbs = generate_limbs(b, L.modulus_memory, L.is_real)
    byte_sum = m_limbs[0]
    for v in m_limbs[1:]:
        byte_sum = byte_sum + v
    eval_is_zero(b, byte_sum, L.modulus_is_zero, L.is_real)
    mz = L.modulus_is_zero.result
    # carry * (the modulus, or 2^256 when it is zero) = carry * (1 - mz) * m(x) + carry * mz x^32
    neg_carry = [-v for v in L.output.carry]
    eval_field_op_polynomials(b, L.output, [], None, L.output.result, L.is_real,
                              products=[(x_limbs, y_limbs), (neg_carry, [1 - mz], m_limbs), (neg_carry, [0] * 32 + [mz])])
    eval_field_lt(b, L.output_range_check, L.output.result, m_limbs, L.modulus_is_not_zero)
    b.assert_eq(L.modulus_is_not_zero, L.is_real * (1 - mz))
    result_words = limbs_to_words(L.output.result)
    x_ptr = eval_syscall_addr(b, 32, L.x_ptr, L.is_real)
    y_ptr = eval_syscall_addr(b, 64, L.y_ptr, L.is_real)
    for i in range(4):
        eval_addr_add(b, list(x_ptr) + [b.const(0)], word_of_u64(8 * i), L.x_addrs[i].value, L.is_real)
    for i in range(8):
        eval_addr_add(b, list(y_ptr) + [b.const(0)], word_of_u64(8 * i), L.y_and_modulus_addrs[i].value, L.is_real)
    for i in range(4):
        eval_memory_access(b, L.clk_high, L.clk_low + 1, L.x_addrs[i].value, L.x_memory[i].memory_access, result_words[i], L.is_real)
    for i, acc in enumerate(L.y_memory + L.modulus_memory):
        eval_memory_access(b, L.clk_high, L.clk_low, L.y_and_modulus_addrs[i].value, acc.memory_access, acc.memory_access.prev_value, L.is_real)
    send_syscall(b, L.clk_high, L.clk_low, SYS_UINT256_MUL, x_ptr, y_ptr, L.is_real, receive=True)
    return _done(b, c)


SECP256K1_P = (1 << 256) - (1 << 32) - 977                                                # curves/src/weierstrass/secp256k1.rs:L29-L45
SYS_SECP256K1_ADD, SYS_SECP256K1_DOUBLE = 0x0A, 0x0B


def eval_field_op(b, cols, a, bb, op, modulus, is_real, witness_offset=WITNESS_OFFSET, result=None):   # FieldOpCols::eval (field_op.rs:L472-L500)
    """result = a op bb mod `modulus` (an integer: the curve's base field) on byte limbs (as many as the `result` columns); sub /
    div are the add / mul identities with the result in a's place (result + bb = a, result * bb = a). `result`: other limbs
    standing in for the result columns (FieldSqrtCols checks sqrt * sqrt = a that way, field_sqrt.rs:L86-L90)."""
    p_mod = [(modulus >> (8 * i)) & 0xFF for i in range(len(cols.result))]
    if result is not None:
        cols = types.SimpleNamespace(result=result, carry=cols.carry, witness=cols.witness)
    if op in ("add", "mul"):
        p_a, p_res = a, cols.result
    else:
        p_a, p_res = cols.result, a
    if op in ("add", "sub"):
        eval_field_op_polynomials(b, cols, _poly_add(p_a, bb), p_mod, p_res, is_real, witness_offset, products=[])
    else:
        eval_field_op_polynomials(b, cols, [], p_mod, p_res, is_real, witness_offset, products=[(p_a, bb)])


# curve: (base-field modulus, a coefficient, byte limbs, witness limbs, witness offset) — curves/src/weierstrass/{secp256k1,secp256r1,bn254,bls12_381}.rs
SECP256R1_P = (1 << 256) - (1 << 224) + (1 << 192) + (1 << 96) - 1
BN254_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BLS12381_P = 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787
CURVES = {"Secp256k1": (SECP256K1_P, 0, 32, 62, 1 << 14), "Secp256r1": (SECP256R1_P, SECP256R1_P - 3, 32, 62, 1 << 14),
          "Bn254": (BN254_P, 0, 32, 62, 1 << 14), "Bls12381": (BLS12381_P, 0, 48, 94, 1 << 15)}
# (SyscallCode's low byte of the curve's ADD, of its DOUBLE) — syscall_code.rs:L74-L159
CURVE_SYSCALLS = {"Secp256k1": (0x0A, 0x0B), "Secp256r1": (0x2C, 0x2D), "Bn254": (0x0E, 0x0F), "Bls12381": (0x1E, 0x1F)}


def weierstrass_add_chip(curve="Secp256k1"):                                              # weierstrass/weierstrass_add.rs:L420-L610
    modulus, _, nl, nw, off = CURVES[curve]
    words = nl // 4                                                                       # u64 words of an affine point
    b, c, _ = _chip(curve + "AddAssign", {32: 1599, 48: 2399}[nl])
    accs = lambda n: (lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(n)])
    addrs = lambda n: (lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(n)])
    FO = FIELD_OP(nl, nw)
    L = S(("is_real", 1), ("clk_high", 1), ("clk_low", 1), ("p_ptr", SYSCALL_ADDR), ("q_ptr", SYSCALL_ADDR), ("p_addrs", addrs(words)), ("q_addrs", addrs(words)),
          ("p_access", accs(words)), ("q_access", accs(words)), ("slope_denominator", FO), ("inverse_check", FO), ("slope_numerator", FO), ("slope", FO),
          ("slope_squared", FO), ("p_x_plus_q_x", FO), ("x3_ins", FO), ("p_x_minus_x", FO), ("y3_ins", FO), ("slope_times_p_x_minus_x", FO),
          ("x3_range", FIELD_LT(nl)), ("y3_range", FIELD_LT(nl)))(c)
    r = L.is_real
    op = lambda cols, x, y, o: eval_field_op(b, cols, x, y, o, modulus, r, off)
    h = words // 2
    p_x, p_y = generate_limbs(b, L.p_access[:h], r), generate_limbs(b, L.p_access[h:], r)
    q_x, q_y = generate_limbs(b, L.q_access[:h], r), generate_limbs(b, L.q_access[h:], r)
    op(L.slope_numerator, q_y, p_y, "sub")
    op(L.slope_denominator, q_x, p_x, "sub")
    one = [b.const(1)] + [b.const(0)] * (nl - 1)
    op(L.inverse_check, one, L.slope_denominator.result, "div")
    op(L.slope, L.slope_numerator.result, L.slope_denominator.result, "div")
    slope = L.slope.result
    op(L.slope_squared, slope, slope, "mul")
    op(L.p_x_plus_q_x, p_x, q_x, "add")
    op(L.x3_ins, L.slope_squared.result, L.p_x_plus_q_x.result, "sub")
    x = L.x3_ins.result
    op(L.p_x_minus_x, p_x, x, "sub")
    op(L.slope_times_p_x_minus_x, slope, L.p_x_minus_x.result, "mul")
    op(L.y3_ins, L.slope_times_p_x_minus_x.result, p_y, "sub")
    mod_limbs = [b.const((modulus >> (8 * i)) & 0xFF) for i in range(nl)]
    eval_field_lt(b, L.x3_range, L.x3_ins.result, mod_limbs, r)
    eval_field_lt(b, L.y3_range, L.y3_ins.result, mod_limbs, r)
    result_words = limbs_to_words(L.x3_ins.result) + limbs_to_words(L.y3_ins.result)
    p_ptr = eval_syscall_addr(b, 2 * nl, L.p_ptr, r)
    q_ptr = eval_syscall_addr(b, 2 * nl, L.q_ptr, r)
    for i in range(words):
        eval_addr_add(b, list(p_ptr) + [b.const(0)], word_of_u64(8 * i), L.p_addrs[i].value, r)
    for i in range(words):
        eval_addr_add(b, list(q_ptr) + [b.const(0)], word_of_u64(8 * i), L.q_addrs[i].value, r)
    for i in range(words):
        acc = L.q_access[i].memory_access
        eval_memory_access(b, L.clk_high, L.clk_low, L.q_addrs[i].value, acc, acc.prev_value, r)
    for i in range(words):
        eval_memory_access(b, L.clk_high, L.clk_low + 1, L.p_addrs[i].value, L.p_access[i].memory_access, result_words[i], r)
    send_syscall(b, L.clk_high, L.clk_low, CURVE_SYSCALLS[curve][0], p_ptr, q_ptr, r, receive=True)
    return _done(b, c)


def weierstrass_double_chip(curve="Secp256k1"):                                           # weierstrass_double.rs:L415-L620
    modulus, a_coeff, nl, nw, off = CURVES[curve]
    words = nl // 4
    b, c, _ = _chip(curve + "DoubleAssign", {32: 1591, 48: 2391}[nl])
    FO = FIELD_OP(nl, nw)
    L = S(("is_real", 1), ("clk_high", 1), ("clk_low", 1), ("p_ptr", SYSCALL_ADDR), ("p_addrs", lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(words)]),
          ("p_access", lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(words)]), ("slope_denominator", FO), ("slope_numerator", FO), ("slope", FO),
          ("p_x_squared", FO), ("p_x_squared_times_3", FO), ("slope_squared", FO), ("p_x_plus_p_x", FO), ("x3_ins", FO), ("p_x_minus_x", FO), ("y3_ins", FO),
          ("slope_times_p_x_minus_x", FO), ("x3_range", FIELD_LT(nl)), ("y3_range", FIELD_LT(nl)))(c)
    r = L.is_real
    op = lambda cols, x, y, o: eval_field_op(b, cols, x, y, o, modulus, r, off)
    h = words // 2
    p_x, p_y = generate_limbs(b, L.p_access[:h], r), generate_limbs(b, L.p_access[h:], r)
    const_limbs = lambda v: [b.const((v >> (8 * i)) & 0xFF) for i in range(nl)]
    op(L.p_x_squared, p_x, p_x, "mul")
    op(L.p_x_squared_times_3, L.p_x_squared.result, const_limbs(3), "mul")
    op(L.slope_numerator, const_limbs(a_coeff), L.p_x_squared_times_3.result, "add")
    op(L.slope_denominator, const_limbs(2), p_y, "mul")
    op(L.slope, L.slope_numerator.result, L.slope_denominator.result, "div")
    slope = L.slope.result
    op(L.slope_squared, slope, slope, "mul")
    op(L.p_x_plus_p_x, p_x, p_x, "add")
    op(L.x3_ins, L.slope_squared.result, L.p_x_plus_p_x.result, "sub")
    op(L.p_x_minus_x, p_x, L.x3_ins.result, "sub")
    op(L.slope_times_p_x_minus_x, slope, L.p_x_minus_x.result, "mul")
    op(L.y3_ins, L.slope_times_p_x_minus_x.result, p_y, "sub")
    eval_field_lt(b, L.x3_range, L.x3_ins.result, const_limbs(modulus), r)
    eval_field_lt(b, L.y3_range, L.y3_ins.result, const_limbs(modulus), r)
    result_words = limbs_to_words(L.x3_ins.result) + limbs_to_words(L.y3_ins.result)
    p_ptr = eval_syscall_addr(b, 2 * nl, L.p_ptr, r)
    for i in range(words):
        eval_addr_add(b, list(p_ptr) + [b.const(0)], word_of_u64(8 * i), L.p_addrs[i].value, r)
    for i in range(words):
        eval_memory_access(b, L.clk_high, L.clk_low, L.p_addrs[i].value, L.p_access[i].memory_access, result_words[i], r)
    send_syscall(b, L.clk_high, L.clk_low, CURVE_SYSCALLS[curve][1], p_ptr, [0, 0, 0], r, receive=True)
    return _done(b, c)


# field-tower precompiles (syscall/precompiles/fptower/): (ADD, SUB, MUL) of Fp, then of Fp2 — syscall_code.rs:L118-L153
FP_FIELDS = {"Bn254": (BN254_P, 32, 62, 1 << 14), "Bls12381": (BLS12381_P, 48, 94, 1 << 15)}
FP_SYSCALLS = {"Bn254": (0x26, 0x27, 0x28, 0x29, 0x2A, 0x2B), "Bls12381": (0x20, 0x21, 0x22, 0x23, 0x24, 0x25)}


def _binary_memory(b, L, words, length):
    """What every x <- x op y precompile does with its two operands: pointer checks, word addresses, y read at clk, x rewritten
    with `result_words` at clk + 1 (fp.rs:L378-L425 and the same lines of the other chips). Returns a closure taking the words."""
    r = L.is_real
    x_ptr = eval_syscall_addr(b, length, L.x_ptr, r)
    y_ptr = eval_syscall_addr(b, length, L.y_ptr, r)
    for i in range(words):
        eval_addr_add(b, list(x_ptr) + [b.const(0)], word_of_u64(8 * i), L.x_addrs[i].value, r)
    for i in range(words):
        eval_addr_add(b, list(y_ptr) + [b.const(0)], word_of_u64(8 * i), L.y_addrs[i].value, r)

    def finish(result_words, syscall_id):
        for i in range(words):
            acc = L.y_access[i].memory_access
            eval_memory_access(b, L.clk_high, L.clk_low, L.y_addrs[i].value, acc, acc.prev_value, r)
        for i in range(words):
            eval_memory_access(b, L.clk_high, L.clk_low + 1, L.x_addrs[i].value, L.x_access[i].memory_access, result_words[i], r)
        send_syscall(b, L.clk_high, L.clk_low, syscall_id, x_ptr, y_ptr, r, receive=True)
    return finish


def eval_field_op_variable(b, cols, a, bb, modulus, is_add, is_sub, is_mul, is_real, witness_offset):   # FieldOpCols::eval_variable (field_op.rs:L367-L401, is_div = 0)
    """p_op - p_result with p_op = is_add (a + b) + is_sub (result + b) + is_mul (a b) and p_result = (is_add + is_mul) result +
    is_sub a, selector by selector: is_add (a + b - result) + is_sub (result + b - a) + is_mul (a b) - is_mul result."""
    p_mod = [(modulus >> (8 * i)) & 0xFF for i in range(len(cols.result))]
    res = cols.result
    terms = [([is_add], _poly_sub(_poly_add(a, bb), res)), ([is_sub], _poly_sub(_poly_add(res, bb), a)), ([is_mul], a, bb), ([is_mul], [-v for v in res])]
    eval_field_op_polynomials(b, cols, [], p_mod, [], is_real, witness_offset, products=terms)


def fp_op_chip(field):                                                                    # fptower/fp.rs:L292-L461
    modulus, nl, nw, off = FP_FIELDS[field]
    words = nl // 8
    b, c, _ = _chip(field + "FpOpAssign", {32: 306, 48: 450}[nl])
    accs = lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(words)]
    addrs = lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(words)]
    L = S(("is_real", 1), ("clk_high", 1), ("clk_low", 1), ("is_add", 1), ("is_sub", 1), ("is_mul", 1), ("x_ptr", SYSCALL_ADDR), ("y_ptr", SYSCALL_ADDR),
          ("x_addrs", addrs), ("y_addrs", addrs), ("x_access", accs), ("y_access", accs), ("output", FIELD_OP(nl, nw)), ("output_range", FIELD_LT(nl)))(c)
    for f in (L.is_add, L.is_sub, L.is_mul, L.is_real):
        b.assert_bool(f)
    b.assert_eq(L.is_add + L.is_sub + L.is_mul, 1)
    p_ = generate_limbs(b, L.x_access, L.is_real)
    q_ = generate_limbs(b, L.y_access, L.is_real)
    eval_field_op_variable(b, L.output, p_, q_, modulus, L.is_add, L.is_sub, L.is_mul, L.is_real, off)
    eval_field_lt(b, L.output_range, L.output.result, [b.const((modulus >> (8 * i)) & 0xFF) for i in range(nl)], L.is_real)
    finish = _binary_memory(b, L, words, nl)
    add_id, sub_id, mul_id = FP_SYSCALLS[field][:3]
    finish(limbs_to_words(L.output.result), L.is_add * add_id + L.is_sub * sub_id + L.is_mul * mul_id)
    return _done(b, c)


def fp2_addsub_chip(field):                                                               # fptower/fp2_addsub.rs:L318-L501
    modulus, nl, nw, off = FP_FIELDS[field]
    words = nl // 4
    b, c, _ = _chip(field + "Fp2AddSubAssign", {32: 592, 48: 880}[nl])
    accs = lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(words)]
    addrs = lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(words)]
    L = S(("is_real", 1), ("clk_high", 1), ("clk_low", 1), ("is_add", 1), ("x_ptr", SYSCALL_ADDR), ("y_ptr", SYSCALL_ADDR), ("x_addrs", addrs), ("y_addrs", addrs),
          ("x_access", accs), ("y_access", accs), ("c0", FIELD_OP(nl, nw)), ("c1", FIELD_OP(nl, nw)), ("c0_range", FIELD_LT(nl)), ("c1_range", FIELD_LT(nl)))(c)
    b.assert_bool(L.is_add)
    r, h = L.is_real, words // 2
    p_x, q_x = generate_limbs(b, L.x_access[:h], r), generate_limbs(b, L.y_access[:h], r)
    p_y, q_y = generate_limbs(b, L.x_access[h:], r), generate_limbs(b, L.y_access[h:], r)
    eval_field_op_variable(b, L.c0, p_x, q_x, modulus, L.is_add, 1 - L.is_add, 0, r, off)
    eval_field_op_variable(b, L.c1, p_y, q_y, modulus, L.is_add, 1 - L.is_add, 0, r, off)
    mod_limbs = [b.const((modulus >> (8 * i)) & 0xFF) for i in range(nl)]
    eval_field_lt(b, L.c0_range, L.c0.result, mod_limbs, r)
    eval_field_lt(b, L.c1_range, L.c1.result, mod_limbs, r)
    finish = _binary_memory(b, L, words, 2 * nl)
    add_id, sub_id = FP_SYSCALLS[field][3:5]
    finish(limbs_to_words(L.c0.result) + limbs_to_words(L.c1.result), L.is_add * add_id + (1 - L.is_add) * sub_id)
    return _done(b, c)


def fp2_mul_chip(field):                                                                  # fptower/fp2_mul.rs:L340-L545
    modulus, nl, nw, off = FP_FIELDS[field]
    words = nl // 4
    b, c, _ = _chip(field + "Fp2MulAssign", {32: 1095, 48: 1639}[nl])
    accs = lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(words)]
    addrs = lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(words)]
    FO = FIELD_OP(nl, nw)
    L = S(("is_real", 1), ("clk_high", 1), ("clk_low", 1), ("x_ptr", SYSCALL_ADDR), ("y_ptr", SYSCALL_ADDR), ("x_addrs", addrs), ("y_addrs", addrs),
          ("x_access", accs), ("y_access", accs), ("a0_mul_b0", FO), ("a1_mul_b1", FO), ("a0_mul_b1", FO), ("a1_mul_b0", FO), ("c0", FO), ("c1", FO),
          ("c0_range", FIELD_LT(nl)), ("c1_range", FIELD_LT(nl)))(c)
    r, h = L.is_real, words // 2
    op = lambda cols, x, y, o: eval_field_op(b, cols, x, y, o, modulus, r, off)
    p_x, q_x = generate_limbs(b, L.x_access[:h], r), generate_limbs(b, L.y_access[:h], r)
    p_y, q_y = generate_limbs(b, L.x_access[h:], r), generate_limbs(b, L.y_access[h:], r)
    op(L.a0_mul_b0, p_x, q_x, "mul")
    op(L.a1_mul_b1, p_y, q_y, "mul")
    op(L.c0, L.a0_mul_b0.result, L.a1_mul_b1.result, "sub")
    op(L.a0_mul_b1, p_x, q_y, "mul")
    op(L.a1_mul_b0, p_y, q_x, "mul")
    op(L.c1, L.a0_mul_b1.result, L.a1_mul_b0.result, "add")
    mod_limbs = [b.const((modulus >> (8 * i)) & 0xFF) for i in range(nl)]
    eval_field_lt(b, L.c0_range, L.c0.result, mod_limbs, r)
    eval_field_lt(b, L.c1_range, L.c1.result, mod_limbs, r)
    finish = _binary_memory(b, L, words, 2 * nl)
    finish(limbs_to_words(L.c0.result) + limbs_to_words(L.c1.result), FP_SYSCALLS[field][5])
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
# ed25519 (syscall/precompiles/edwards/; curves/src/edwards/ed25519.rs:L24-L50)
ED25519_P = (1 << 255) - 19
ED25519_D = 37095705934669439343138083508754565189542113879843219016388785533085940283555
SYS_ED_ADD, SYS_ED_DECOMPRESS = 0x07, 0x08


def eval_field_inner_product(b, cols, a_list, b_list, modulus, is_real):                  # field_inner_product.rs:L106-L145
    p_mod = [(modulus >> (8 * i)) & 0xFF for i in range(len(cols.result))]
    eval_field_op_polynomials(b, cols, [], p_mod, cols.result, is_real, products=list(zip(a_list, b_list)))


def eval_field_den(b, cols, a, bb, sign, modulus, is_real):                               # field_den.rs:L113-L150: result = a / (1 + bb) or a / (1 - bb)
    p_mod = [(modulus >> (8 * i)) & 0xFF for i in range(len(cols.result))]
    eval_field_op_polynomials(b, cols, cols.result if sign else a, p_mod, a if sign else cols.result, is_real, products=[(bb, cols.result)])


def ed_add_chip():                                                                        # edwards/ed_add.rs:L330-L470
    b, c, _ = _chip("EdAddAssign", 1347)
    accs = lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(8)]
    addrs = lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(8)]
    FO = FIELD_OP(32, 62)
    L = S(("is_real", 1), ("clk_high", 1), ("clk_low", 1), ("x_ptr", SYSCALL_ADDR), ("y_ptr", SYSCALL_ADDR), ("x_addrs", addrs), ("y_addrs", addrs),
          ("x_access", accs), ("y_access", accs), ("x3_numerator", FO), ("y3_numerator", FO), ("x1_mul_y1", FO), ("x2_mul_y2", FO), ("f", FO), ("d_mul_f", FO),
          ("x3_ins", FO), ("y3_ins", FO), ("x3_range", FIELD_LT(32)), ("y3_range", FIELD_LT(32)))(c)            # x_* = the reference's p_*, y_* = its q_*
    r = L.is_real
    x1, x2 = generate_limbs(b, L.x_access[:4], r), generate_limbs(b, L.y_access[:4], r)
    y1, y2 = generate_limbs(b, L.x_access[4:], r), generate_limbs(b, L.y_access[4:], r)
    eval_field_inner_product(b, L.x3_numerator, [x1, x2], [y2, y1], ED25519_P, r)
    eval_field_inner_product(b, L.y3_numerator, [y1, x1], [y2, x2], ED25519_P, r)
    op = lambda cols, x, y, o: eval_field_op(b, cols, x, y, o, ED25519_P, r)
    op(L.x1_mul_y1, x1, y1, "mul")
    op(L.x2_mul_y2, x2, y2, "mul")
    op(L.f, L.x1_mul_y1.result, L.x2_mul_y2.result, "mul")
    const_limbs = lambda v: [b.const((v >> (8 * i)) & 0xFF) for i in range(32)]
    op(L.d_mul_f, L.f.result, const_limbs(ED25519_D), "mul")
    eval_field_den(b, L.x3_ins, L.x3_numerator.result, L.d_mul_f.result, True, ED25519_P, r)
    eval_field_lt(b, L.x3_range, L.x3_ins.result, const_limbs(ED25519_P), r)
    eval_field_den(b, L.y3_ins, L.y3_numerator.result, L.d_mul_f.result, False, ED25519_P, r)
    eval_field_lt(b, L.y3_range, L.y3_ins.result, const_limbs(ED25519_P), r)
    result_words = limbs_to_words(L.x3_ins.result) + limbs_to_words(L.y3_ins.result)
    x_ptr = eval_syscall_addr(b, 64, L.x_ptr, r)
    y_ptr = eval_syscall_addr(b, 64, L.y_ptr, r)
    for i in range(8):
        eval_addr_add(b, list(y_ptr) + [b.const(0)], word_of_u64(8 * i), L.y_addrs[i].value, r)
    for i in range(8):
        eval_addr_add(b, list(x_ptr) + [b.const(0)], word_of_u64(8 * i), L.x_addrs[i].value, r)
    for i in range(8):
        acc = L.y_access[i].memory_access
        eval_memory_access(b, L.clk_high, L.clk_low, L.y_addrs[i].value, acc, acc.prev_value, r)
    for i in range(8):
        eval_memory_access(b, L.clk_high, L.clk_low + 1, L.x_addrs[i].value, L.x_access[i].memory_access, result_words[i], r)
    send_syscall(b, L.clk_high, L.clk_low, SYS_ED_ADD, x_ptr, y_ptr, r, receive=True)
    return _done(b, c)


def ed_decompress_chip():                                                                 # edwards/ed_decompress.rs:L197-L330, L451-L476
    b, c, _ = _chip("EdDecompress", 1123)
    FO = FIELD_OP(32, 62)
    L = S(("is_real", 1), ("clk_high", 1), ("clk_low", 1), ("ptr", SYSCALL_ADDR), ("read_ptrs", lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(4)]),
          ("addrs", lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(4)]), ("sign", 1),
          ("x_access", lambda c_, p: [MEM_ACCESS(c_, p + "%d." % i) for i in range(4)]), ("x_value", lambda c_, p: [c_.arr(4, p + "%d" % i) for i in range(4)]),
          ("y_access", lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(4)]), ("neg_x_range", FIELD_LT(32)), ("y_range", FIELD_LT(32)),
          ("yy", FO), ("u", FO), ("dyy", FO), ("v", FO), ("u_div_v", FO),
          ("x", S(("multiplication", FO), ("range", FIELD_LT(32)), ("lsb", 1))), ("neg_x", FO))(c)
    r = L.is_real
    b.assert_bool(L.sign)
    b.assert_bool(r)
    y = generate_limbs(b, L.y_access, r)
    const_limbs = lambda v: [b.const((v >> (8 * i)) & 0xFF) for i in range(32)]
    mod_limbs = const_limbs(ED25519_P)
    op = lambda cols, x, yv, o, **kw: eval_field_op(b, cols, x, yv, o, ED25519_P, r, **kw)
    eval_field_lt(b, L.y_range, y, mod_limbs, r)
    op(L.yy, y, y, "mul")
    op(L.u, L.yy.result, [b.const(1)], "sub")
    op(L.dyy, const_limbs(ED25519_D), L.yy.result, "mul")
    op(L.v, [b.const(1)], L.dyy.result, "add")
    op(L.u_div_v, L.u.result, L.v.result, "div")
    sqrt = L.x.multiplication.result                                                      # FieldSqrtCols::eval (field_sqrt.rs:L75-L113), is_odd = 0
    op(L.x.multiplication, sqrt, sqrt, "mul", result=L.u_div_v.result)
    eval_field_lt(b, L.x.range, sqrt, mod_limbs, r)
    slice_range_check_u8(b, sqrt, r)
    b.assert_bool(L.x.lsb)
    b.when(r).assert_eq(L.x.lsb, 0)
    send_byte(b, B_AND, L.x.lsb, sqrt[0], 1, r)
    op(L.neg_x, [b.const(0)], sqrt, "sub")
    eval_field_lt(b, L.neg_x_range, L.neg_x.result, mod_limbs, r)
    ptr = eval_syscall_addr(b, 64, L.ptr, r)
    for i in range(4):
        eval_addr_add(b, list(ptr) + [b.const(0)], word_of_u64(8 * i), L.addrs[i].value, r)
    for i in range(4):
        eval_addr_add(b, list(ptr) + [b.const(0)], word_of_u64(8 * i + 32), L.read_ptrs[i].value, r)
    for i in range(4):
        acc = L.y_access[i].memory_access
        eval_memory_access(b, L.clk_high, L.clk_low, L.read_ptrs[i].value, acc, acc.prev_value, r)
    for i in range(4):
        eval_memory_access(b, L.clk_high, L.clk_low + 1, L.addrs[i].value, L.x_access[i], L.x_value[i], r)
    for words, cond in ((limbs_to_words(L.neg_x.result), L.sign), (limbs_to_words(sqrt), 1 - L.sign)):
        for w, xv in zip(words, L.x_value):
            for k in range(4):
                b.when(r).when(cond).assert_eq(w[k], xv[k])
    send_syscall(b, L.clk_high, L.clk_low, SYS_ED_DECOMPRESS, ptr, [L.sign, 0, 0], r, receive=True)
    return _done(b, c)


# ---------------------------------------------------------------------------------------------------------------------
SYS_UINT256_ADD_CARRY, SYS_UINT256_MUL_CARRY = 0x30, 0x31


def uint256_ops_chip():                                                                   # syscall/precompiles/uint256_ops/air.rs:L117-L402
    """d, e <- low and high 256 bits of a + b + c or a * b + c: a and b at the call's two arguments, the pointers c, d, e in
    registers x12, x13, x14 (read by this chip); five slices at clk .. clk + 4."""
    b, c, _ = _chip("Uint256Ops", 477)
    addrs = lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(4)]
    accs8 = lambda c_, p: [MEM_ACCESS_U8(c_, p + "%d." % i) for i in range(4)]
    accs = lambda c_, p: [MEM_ACCESS(c_, p + "%d." % i) for i in range(4)]
    L = S(("clk_high", 1), ("clk_low", 1), ("a_ptr", SYSCALL_ADDR), ("a_addrs", addrs), ("b_ptr", SYSCALL_ADDR), ("b_addrs", addrs),
          ("c_ptr", SYSCALL_ADDR), ("c_ptr_memory", MEM_ACCESS), ("c_addrs", addrs), ("d_ptr", SYSCALL_ADDR), ("d_ptr_memory", MEM_ACCESS), ("d_addrs", addrs),
          ("e_ptr", SYSCALL_ADDR), ("e_ptr_memory", MEM_ACCESS), ("e_addrs", addrs), ("a_memory", accs8), ("b_memory", accs8), ("c_memory", accs8),
          ("d_memory", accs), ("e_memory", accs), ("field_op", FIELD_OP(32, 63)), ("is_add", 1), ("is_mul", 1), ("is_real", 1))(c)
    r = L.is_real
    b.assert_bool(L.is_add)
    b.assert_bool(L.is_mul)
    b.assert_bool(r)
    b.assert_eq(r, L.is_add + L.is_mul)
    ptrs = {k: eval_syscall_addr(b, 32, getattr(L, k + "_ptr"), r) for k in "abcde"}
    send_syscall(b, L.clk_high, L.clk_low, L.is_add * SYS_UINT256_ADD_CARRY + L.is_mul * SYS_UINT256_MUL_CARRY, ptrs["a"], ptrs["b"], r, receive=True)
    for k, reg in (("c", 12), ("d", 13), ("e", 14)):
        acc = getattr(L, k + "_ptr_memory")
        eval_memory_access(b, L.clk_high, L.clk_low, [b.const(reg), b.const(0), b.const(0)], acc, acc.prev_value, r)
        for have, want in zip(acc.prev_value, list(ptrs[k]) + [b.const(0)]):
            b.assert_eq(have, want)
    for i in range(4):
        for k in "abcde":
            eval_addr_add(b, list(ptrs[k]) + [b.const(0)], word_of_u64(8 * i), getattr(L, k + "_addrs")[i].value, r)
    for at, k in enumerate("abc"):
        for i in range(4):
            acc = getattr(L, k + "_memory")[i].memory_access
            eval_memory_access(b, L.clk_high, L.clk_low + at, getattr(L, k + "_addrs")[i].value, acc, acc.prev_value, r)
    a_l, b_l, c_l = (generate_limbs(b, getattr(L, k + "_memory"), r) for k in "abc")
    # eval_add_mul_and_carry: p_op = is_add (a + b) + is_mul (a b) + c
    eval_field_op_polynomials(b, L.field_op, c_l, [0] * 32 + [1], L.field_op.result, r, products=[([L.is_add], _poly_add(a_l, b_l)), ([L.is_mul], a_l, b_l)])
    for at, k, limbs in ((3, "d", L.field_op.result), (4, "e", L.field_op.carry)):
        words = limbs_to_words(limbs)
        for i in range(4):
            eval_memory_access(b, L.clk_high, L.clk_low + at, getattr(L, k + "_addrs")[i].value, getattr(L, k + "_memory")[i], words[i], r)
    return _done(b, c)


def poseidon2_chip():                                                                     # syscall/precompiles/poseidon2/air.rs:L424-L607
    """The POSEIDON2 precompile: eight u64 words at `ptr` (sixteen field elements, low half first) are read and rewritten in place
    by one KoalaBear Poseidon2 permutation — the same `Poseidon2Operation` sub-AIR as the Global chip's (hinted for a fused kernel)."""
    from .recursion import P2_EXT, P2_OUT, P2_WIDTH, poseidon2_permutation_constraints
    b, c, _ = _chip("Poseidon2", 348)
    L = S(("clk_high", 1), ("clk_low", 1), ("ptr", SYSCALL_ADDR), ("addrs", lambda c_, p: [ADDR_ADD_OP(c_, p + "%d." % i) for i in range(8)]),
          ("memory", lambda c_, p: [MEM_ACCESS(c_, p + "%d." % i) for i in range(8)]),
          ("hash_result", lambda c_, p: [c_.arr(4, p + "%d" % i) for i in range(8)]), ("hash_result_range_checkers", 16),
          ("input_range_checkers", 16), ("permutation", P2_WIDTH), ("is_real", 1))(c)
    ptr = eval_syscall_addr(b, 64, L.ptr, L.is_real)
    for i in range(8):
        eval_addr_add(b, list(ptr) + [b.const(0)], word_of_u64(8 * i), L.addrs[i].value, L.is_real)
    for i in range(8):                                                                    # eval_memory_access_slice_write (air/memory.rs)
        eval_memory_access(b, L.clk_high, L.clk_low, L.addrs[i].value, L.memory[i], L.hash_result[i], L.is_real)
    inputs, outputs = [], []
    for words, checkers, out in (([m.prev_value for m in L.memory], L.input_range_checkers, inputs),
                                 (L.hash_result, L.hash_result_range_checkers, outputs)):
        for i in range(8):
            w = words[i]
            out += [w[0] + w[1] * (1 << 16), w[2] + w[3] * (1 << 16)]
            slice_range_check_u16(b, w, L.is_real)
            eval_field_word_range_check(b, [w[0], w[1], b.const(0), b.const(0)], checkers[2 * i], L.is_real)
            eval_field_word_range_check(b, [w[2], w[3], b.const(0), b.const(0)], checkers[2 * i + 1], L.is_real)
    perm = L.permutation
    for i in range(16):
        b.when(L.is_real).assert_eq(perm[P2_EXT(0, i)], inputs[i])
    poseidon2_permutation_constraints(b.air, c.names["permutation"])
    for i in range(16):
        b.when(L.is_real).assert_eq(perm[P2_OUT(i)], outputs[i])
    send_syscall(b, L.clk_high, L.clk_low, SYS_POSEIDON2, ptr, [0, 0, 0], L.is_real, receive=True)
    b.assert_bool(L.is_real)
    return _done(b, c)


MORE_CHIPS = {
    "Secp256k1AddAssign": weierstrass_add_chip, "Secp256k1DoubleAssign": weierstrass_double_chip,
    **{cv + "AddAssign": (lambda cv=cv: weierstrass_add_chip(cv)) for cv in ("Secp256r1", "Bn254", "Bls12381")},
    **{cv + "DoubleAssign": (lambda cv=cv: weierstrass_double_chip(cv)) for cv in ("Secp256r1", "Bn254", "Bls12381")},
    **{f + "FpOpAssign": (lambda f=f: fp_op_chip(f)) for f in FP_FIELDS}, **{f + "Fp2AddSubAssign": (lambda f=f: fp2_addsub_chip(f)) for f in FP_FIELDS},
    **{f + "Fp2MulAssign": (lambda f=f: fp2_mul_chip(f)) for f in FP_FIELDS}, "EdAddAssign": ed_add_chip, "EdDecompress": ed_decompress_chip,
    "Uint256Ops": uint256_ops_chip, "Uint256MulMod": uint256_mul_chip, "Poseidon2": poseidon2_chip, "ShaExtend": sha_extend_chip, "ShaExtendControl": sha_extend_control_chip, "ShaCompress": sha_compress_chip,
    "ShaCompressControl": sha_compress_control_chip,
    "AluX0": alu_x0_chip, "DivRem": divrem_chip, "SyscallCore": lambda: syscall_chip("core"), "SyscallPrecompile": lambda: syscall_chip("precompile"),
    "SyscallInstrs": syscall_instrs_chip, "MemoryGlobalInit": lambda: memory_global_chip("init"),
    "MemoryGlobalFinalize": lambda: memory_global_chip("finalize"), "KeccakPermute": keccak_permute_chip,
    "KeccakPermuteControl": keccak_control_chip,
}
# (columns, constraints) from rv64im_costs.json / rv64im_complexity.json; interactions of the recorded core shard where it has the chip
MORE_RECORDED = {
    "Secp256k1AddAssign": (1599, 918, None), "Secp256k1DoubleAssign": (1591, 904, None), "Secp256r1AddAssign": (1599, 918, None),
    "Secp256r1DoubleAssign": (1591, 904, None), "Bn254AddAssign": (1599, 918, None), "Bn254DoubleAssign": (1591, 904, None),
    "Bls12381AddAssign": (2399, 1374, None), "Bls12381DoubleAssign": (2391, 1356, None),
    "Bn254FpOpAssign": (306, 217, None), "Bls12381FpOpAssign": (450, 317, None), "Bn254Fp2AddSubAssign": (592, 415, None), "Bls12381Fp2AddSubAssign": (880, 615, None),
    "Bn254Fp2MulAssign": (1095, 666, None), "Bls12381Fp2MulAssign": (1639, 994, None), "EdAddAssign": (1347, 792, None), "EdDecompress": (1123, 755, None),
    "Uint256Ops": (477, 297, None), "Uint256MulMod": (371, 253, None), "Poseidon2": (348, 497, None), "ShaExtend": (128, 80, None), "ShaExtendControl": (18, 21, None), "ShaCompress": (206, 300, None),
    "ShaCompressControl": (53, 21, None), "AluX0": (34, 17, None), "DivRem": (246, 348, 135), "SyscallCore": (10, 2, 4), "SyscallPrecompile": (10, 2, None), "SyscallInstrs": (65, 93, 30),
    "MemoryGlobalInit": (30, 31, None), "MemoryGlobalFinalize": (30, 31, None), "KeccakPermute": (2640, 2859, None),
    "KeccakPermuteControl": (634, 331, None),
}
