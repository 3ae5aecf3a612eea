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
from ..air import P
from .riscv import (ADDRESS_OP, ALU_TYPE, B_LTU, B_RANGE, B_U8RANGE, BYTE, CLK_INC, CPU_STATE, GLOBAL, INV, LT_UNSIGNED, MEM_ACCESS, MEMORY, MUL_OP, OPC,
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
    "Poseidon2": poseidon2_chip,
    "AluX0": alu_x0_chip, "DivRem": divrem_chip, "SyscallCore": lambda: syscall_chip("core"), "SyscallPrecompile": lambda: syscall_chip("precompile"),
    "SyscallInstrs": syscall_instrs_chip, "MemoryGlobalInit": lambda: memory_global_chip("init"),
    "MemoryGlobalFinalize": lambda: memory_global_chip("finalize"), "KeccakPermute": keccak_permute_chip,
    "KeccakPermuteControl": keccak_control_chip,
}
# (columns, constraints) from rv64im_costs.json / rv64im_complexity.json; interactions of the recorded core shard where it has the chip
MORE_RECORDED = {
    "Poseidon2": (348, 497, None), "AluX0": (34, 17, None), "DivRem": (246, 348, 135), "SyscallCore": (10, 2, 4), "SyscallPrecompile": (10, 2, None), "SyscallInstrs": (65, 93, 30),
    "MemoryGlobalInit": (30, 31, None), "MemoryGlobalFinalize": (30, 31, None), "KeccakPermute": (2640, 2859, None),
    "KeccakPermuteControl": (634, 331, None),
}
