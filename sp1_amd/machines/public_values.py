"""The record-level statement of a RISC-V shard: `ExecutionRecord::eval_public_values` as DATA, the layout of `PublicValues`,
how the executor / controller fill it per shard kind, and the cross-shard checks of `SP1Prover::verify`.

    eval_public_values + eval_* helpers      /root/reference/crates/core/executor/src/record.rs:L879-L1510
    PublicValues<[T;4],[T;3],[T;4],T>         /root/reference/crates/hypercube/src/air/public_values.rs:L33-L168 (`mprotect` off: 160 words)
    finalize_public_values                    record.rs:L1596-L1638
    update_initialized_state / finalized      public_values.rs:L219-L345 (precompile shards / memory shards)
    Byte / Range dependencies of the values   crates/core/machine/src/{bytes,range}/trace.rs generate_dependencies
    verify_public_values, the cumulative sum  /root/reference/crates/hypercube/src/logup_gkr/verifier.rs:L74-L96, L138-L181
    the checks across the shards of a proof   /root/reference/crates/prover/src/verify.rs:L106-L520

The program is a pair in the formats the rest of this package uses (sp1_amd/air.py):

  * an `AirProgram` of width 0 whose only loads are PUBLIC words: the `assert_zero`s of eval_public_values in call order — the
    verifier folds them with `pv_challenge` (Horner) and requires zero;
  * an `InteractionProgram` over ONE row whose "main columns" are the 160 public words: the sends / receives of
    eval_public_values (timestamp / pc / address range checks on the Byte bus, the shard's initial and final CPU state on the
    State bus, the two ends of the GlobalAccumulation, MemoryGlobalInitControl and MemoryGlobalFinalizeControl chains, the
    page-protection chains at multiplicity `is_untrusted_programs_enabled`). The verifier evaluates
    sum +-multiplicity / (alpha + beta . (kind, values)) over them: the LogUp-GKR output must sum to minus that.

So a shard's chips alone do NOT balance: what is missing is exactly this row (tests/machine_check.py joins it to the buses).
"""
import torch

from ..air import AirProgram, InteractionProgram, P
from . import riscv as R
from .rv_builder import _PUB, Builder, Sym

NUM_PV_ELTS = 160                   # SP1_PROOF_NUM_PV_ELTS = size_of::<PublicValues<[u8;4],[u8;3],[u8;4],u8>>()
PROOF_MAX_NUM_PVS = 187             # hypercube/src/verifier/proof.rs:L25: what a ShardProof carries (zero beyond NUM_PV_ELTS)
PV_DIGEST_NUM_WORDS, POSEIDON_NUM_WORDS, PROOF_NONCE_NUM_WORDS = 8, 8, 4
HALT_PC = 1                         # core/executor: the pc after HALT

# word offsets, field by field in declaration order (#[repr(C)])
_FIELDS = [("prev_committed_value_digest", 32), ("committed_value_digest", 32), ("prev_deferred_proofs_digest", 8),
           ("deferred_proofs_digest", 8), ("pc_start", 3), ("next_pc", 3), ("prev_exit_code", 1), ("exit_code", 1),
           ("is_execution_shard", 1), ("previous_init_addr", 3), ("last_init_addr", 3), ("previous_finalize_addr", 3),
           ("last_finalize_addr", 3), ("previous_init_page_idx", 3), ("last_init_page_idx", 3), ("previous_finalize_page_idx", 3),
           ("last_finalize_page_idx", 3), ("initial_timestamp", 4), ("last_timestamp", 4), ("is_timestamp_high_eq", 1),
           ("inv_timestamp_high", 1), ("is_timestamp_low_eq", 1), ("inv_timestamp_low", 1), ("global_init_count", 1),
           ("global_finalize_count", 1), ("global_page_prot_init_count", 1), ("global_page_prot_finalize_count", 1),
           ("global_count", 1), ("global_cumulative_sum", 14), ("prev_commit_syscall", 1), ("commit_syscall", 1),
           ("prev_commit_deferred_syscall", 1), ("commit_deferred_syscall", 1), ("initial_timestamp_inv", 1),
           ("last_timestamp_inv", 1), ("is_first_execution_shard", 1), ("is_untrusted_programs_enabled", 1), ("proof_nonce", 4),
           ("empty", 4)]
PV, PV_LEN = {}, {}
_o = 0
for _n, _k in _FIELDS:
    PV[_n], PV_LEN[_n] = _o, _k
    _o += _k
assert _o == NUM_PV_ELTS

# InteractionKind (hypercube/src/lookup/interaction.rs:L27-L80)
MEMORY_GLOBAL_INIT_CONTROL, MEMORY_GLOBAL_FINALIZE_CONTROL, PAGE_PROT_GLOBAL_INIT_CONTROL, PAGE_PROT_GLOBAL_FINALIZE_CONTROL = 14, 15, 20, 21


class _PvBuilder(Builder):
    """The recording builder over public values: `public(i)` lowers to a PUBLIC load in the constraint program and is column i
    of the one-row table the interaction program reads."""

    def __init__(self):
        self.name = "PublicValues"
        self.air = AirProgram(self.name, 0, 0, cse=True)
        self.it = InteractionProgram(self.name, NUM_PV_ELTS, 0)
        self._consts, self._cols = {}, {}

    def public(self, i):
        s = self._cols.get(i)
        if s is None:
            assert 0 <= i < NUM_PV_ELTS
            s = self._cols[i] = Sym(self, _PUB, i, None, ({("main", i): 1}, 0))
        return s

    def field(self, name):
        v = [self.public(PV[name] + i) for i in range(PV_LEN[name])]
        return v[0] if len(v) == 1 else v


def _eval_state(b, pv):                                                    # record.rs:L921-L1087
    it, lt = pv("initial_timestamp"), pv("last_timestamp")
    it_high, it_low = it[1] + it[0] * (1 << 8), it[3] + it[2] * (1 << 16)
    lt_high, lt_low = lt[1] + lt[0] * (1 << 8), lt[3] + lt[2] * (1 << 16)
    inv8 = R.INV(8)
    R.send_byte(b, R.B_RANGE, it[0], 16, 0, 1)
    R.send_byte(b, R.B_RANGE, (it[3] - 1) * inv8, 13, 0, 1)
    R.send_byte(b, R.B_RANGE, lt[0], 16, 0, 1)
    R.send_byte(b, R.B_RANGE, (lt[3] - 1) * inv8, 13, 0, 1)
    R.send_byte(b, R.B_U8RANGE, 0, it[1], it[2], 1)
    R.send_byte(b, R.B_U8RANGE, 0, lt[1], lt[2], 1)
    pc_start, next_pc = pv("pc_start"), pv("next_pc")
    for i in range(3):
        R.send_byte(b, R.B_RANGE, pc_start[i], 16, 0, 1)
        R.send_byte(b, R.B_RANGE, next_pc[i], 16, 0, 1)
    R.send_state(b, it_high, it_low, pc_start, 1)
    R.receive_state(b, lt_high, lt_low, next_pc, 1)
    is_exec = pv("is_execution_shard")
    b.assert_bool(is_exec)
    b.when_not(is_exec).assert_eq(it_low, lt_low)
    b.when_not(is_exec).assert_eq(it_high, lt_high)
    b.when_not(is_exec).assert_all_eq(pc_start, next_pc)
    high_eq, low_eq = pv("is_timestamp_high_eq"), pv("is_timestamp_low_eq")
    b.assert_bool(high_eq)                                                 # IsZeroOperation on the high bits
    b.assert_eq((lt_high - it_high) * pv("inv_timestamp_high"), 1 - high_eq)
    b.assert_zero((lt_high - it_high) * high_eq)
    b.assert_bool(low_eq)                                                  # ... and on the low bits
    b.assert_eq((lt_low - it_low) * pv("inv_timestamp_low"), 1 - low_eq)
    b.assert_zero((lt_low - it_low) * low_eq)
    b.assert_eq(1 - is_exec, high_eq * low_eq)                             # an execution shard's timestamp moves
    b.when(is_exec).assert_eq((lt_high + lt_low - 1) * pv("last_timestamp_inv"), 1)     # ... and does not end at 1


def _eval_first_execution_shard(b, pv):                                    # record.rs:L1089-L1168
    first = pv("is_first_execution_shard")
    b.assert_bool(first)
    b.when(first).assert_all_eq(pv("initial_timestamp"), [0, 0, 0, 1])
    b.when(first).assert_one(pv("is_execution_shard"))
    prev = pv("prev_committed_value_digest")
    for i in range(PV_DIGEST_NUM_WORDS):
        b.when(first).assert_word_zero(prev[4 * i:4 * i + 4])
    b.when(first).assert_word_zero(pv("prev_deferred_proofs_digest"))
    b.when(first).assert_zero(pv("prev_exit_code"))
    for name in ("previous_init_addr", "previous_finalize_addr", "previous_init_page_idx", "previous_finalize_page_idx"):
        b.when(first).assert_word_zero(pv(name))
    b.when(first).assert_zero(pv("prev_commit_syscall"))
    b.when(first).assert_zero(pv("prev_commit_deferred_syscall"))


def _eval_exit_code(b, pv):                                                # record.rs:L1170-L1193
    prev, cur = pv("prev_exit_code"), pv("exit_code")
    b.assert_zero(prev * (cur - prev))
    b.when_not(pv("is_execution_shard")).assert_eq(prev, cur)


def _eval_committed_value_digest(b, pv):                                   # record.rs:L1195-L1274
    is_exec = pv("is_execution_shard")
    prev, cur = pv("prev_committed_value_digest"), pv("committed_value_digest")
    word = lambda d, i: d[4 * i:4 * i + 4]
    for i in range(PV_DIGEST_NUM_WORDS):
        for d in (prev, cur):
            w = word(d, i)
            R.send_byte(b, R.B_U8RANGE, 0, w[0], w[1], 1)
            R.send_byte(b, R.B_U8RANGE, 0, w[2], w[3], 1)
    pcs, cs = pv("prev_commit_syscall"), pv("commit_syscall")
    b.assert_bool(pcs)
    b.assert_bool(cs)
    b.when(pcs).assert_one(cs)
    b.when_not(is_exec).assert_eq(pcs, cs)
    for i in range(PV_DIGEST_NUM_WORDS):
        b.when_not(is_exec).assert_all_eq(word(prev, i), word(cur, i))
    for limb in prev:                                                      # a non-zero previous digest stays
        for i in range(PV_DIGEST_NUM_WORDS):
            b.when(limb).assert_all_eq(word(prev, i), word(cur, i))
    for i in range(PV_DIGEST_NUM_WORDS):
        b.when(pcs).assert_all_eq(word(prev, i), word(cur, i))


def _eval_deferred_proofs_digest(b, pv):                                   # record.rs:L1276-L1327
    is_exec = pv("is_execution_shard")
    prev, cur = pv("prev_deferred_proofs_digest"), pv("deferred_proofs_digest")
    pcs, cs = pv("prev_commit_deferred_syscall"), pv("commit_deferred_syscall")
    b.assert_bool(pcs)
    b.assert_bool(cs)
    b.when(pcs).assert_one(cs)
    b.when_not(is_exec).assert_eq(pcs, cs)
    b.when_not(is_exec).assert_all_eq(prev, cur)
    for limb in prev:
        b.when(limb).assert_all_eq(prev, cur)
    b.when(pcs).assert_all_eq(prev, cur)


def _eval_global_sum(b, pv):                                               # record.rs:L1329-L1363
    x0, y0 = R.CURVE_CUMULATIVE_SUM_START                                  # SepticDigest::zero()
    b.send(R.GLOBAL_ACC, [0] + list(x0) + list(y0), 1)
    b.receive(R.GLOBAL_ACC, [pv("global_count")] + pv("global_cumulative_sum"), 1)


def _eval_address_chain(b, pv, kind, previous, last, count, range_check, mult):
    if range_check:                                                        # record.rs:L1376-L1393, L1428-L1445
        for i in range(3):
            R.send_byte(b, R.B_RANGE, pv(previous)[i], 16, 0, 1)
            R.send_byte(b, R.B_RANGE, pv(last)[i], 16, 0, 1)
    b.send(kind, [0] + pv(previous) + [1], mult)
    b.receive(kind, [pv(count)] + pv(last) + [1], mult)


_PROGRAM = None


def program():
    """(AirProgram, InteractionProgram) of eval_public_values (record.rs:L879-L906), built once."""
    global _PROGRAM
    if _PROGRAM is None:
        b = _PvBuilder()
        pv = b.field
        for var in pv("empty"):
            b.assert_zero(var)
        _eval_state(b, pv)
        _eval_first_execution_shard(b, pv)
        _eval_exit_code(b, pv)
        _eval_committed_value_digest(b, pv)
        _eval_deferred_proofs_digest(b, pv)
        _eval_global_sum(b, pv)
        _eval_address_chain(b, pv, MEMORY_GLOBAL_INIT_CONTROL, "previous_init_addr", "last_init_addr", "global_init_count", True, 1)
        _eval_address_chain(b, pv, MEMORY_GLOBAL_FINALIZE_CONTROL, "previous_finalize_addr", "last_finalize_addr", "global_finalize_count", True, 1)
        untrusted = pv("is_untrusted_programs_enabled")
        b.assert_bool(untrusted)                                           # eval_global_page_prot_init, record.rs:L1467-L1498
        _eval_address_chain(b, pv, PAGE_PROT_GLOBAL_INIT_CONTROL, "previous_init_page_idx", "last_init_page_idx",
                            "global_page_prot_init_count", False, untrusted)
        b.assert_bool(untrusted)                                           # eval_global_page_prot_finalize, L1500-L1531
        _eval_address_chain(b, pv, PAGE_PROT_GLOBAL_FINALIZE_CONTROL, "previous_finalize_page_idx", "last_finalize_page_idx",
                            "global_page_prot_finalize_count", False, untrusted)
        _PROGRAM = (b.air, b.it)
    return _PROGRAM


def max_interaction_arity():
    """`interactions_in_public_values().map(|k| k.num_values() + 1).max()` (verifier.rs:L120-L124): the widest message kind that
    CAN appear in eval_public_values — GlobalAccumulation, 15 values — whatever a given machine build sends."""
    return 16


# ---------------------------------------------------------------------------------------------------------------------
# filling the words
def blank():
    return [0] * NUM_PV_ELTS


def put(pv, name, vals):
    vals = [int(v) for v in (vals if isinstance(vals, (list, tuple)) else [vals])]
    assert len(vals) == PV_LEN[name], (name, len(vals))
    pv[PV[name]:PV[name] + len(vals)] = vals


def get(pv, name):
    v = [int(x) for x in pv[PV[name]:PV[name] + PV_LEN[name]]]
    return v[0] if len(v) == 1 else v


def addr_limbs(a):
    return [(int(a) >> (16 * i)) & 0xFFFF for i in range(3)]


def timestamp_limbs(t):
    """[bits 32..48, bits 24..32, bits 16..24, bits 0..16] (public_values.rs:L641-L652)."""
    t = int(t)
    return [(t >> 32) & 0xFFFF, (t >> 24) & 0xFF, (t >> 16) & 0xFF, t & 0xFFFF]


def timestamp_of(limbs):
    return (int(limbs[0]) << 32) | (int(limbs[1]) << 24) | (int(limbs[2]) << 16) | int(limbs[3])


def digest_bytes(words):
    """committed_value_digest: 8 u32 words -> 32 byte limbs (`u32::to_le_bytes`)."""
    return [(int(w) >> (8 * i)) & 0xFF for w in words for i in range(4)]


def set_state(pv, pc_start, next_pc, initial_timestamp, last_timestamp, exit_code, is_execution_shard):
    """The execution state of a shard + `finalize_public_values` (record.rs:L1596-L1638)."""
    put(pv, "pc_start", addr_limbs(pc_start))
    put(pv, "next_pc", addr_limbs(next_pc))
    put(pv, "initial_timestamp", timestamp_limbs(initial_timestamp))
    put(pv, "last_timestamp", timestamp_limbs(last_timestamp))
    put(pv, "exit_code", exit_code)
    put(pv, "is_execution_shard", int(bool(is_execution_shard)))
    inv = lambda v: pow(v % P, P - 2, P)
    ih, il, lh, ll = initial_timestamp >> 24, initial_timestamp & 0xFFFFFF, last_timestamp >> 24, last_timestamp & 0xFFFFFF
    put(pv, "initial_timestamp_inv", 0 if initial_timestamp == 1 else inv(ih + il - 1))
    put(pv, "last_timestamp_inv", inv(lh + ll - 1))
    put(pv, "is_timestamp_high_eq", int(ih == lh))
    put(pv, "inv_timestamp_high", 0 if ih == lh else inv(lh - ih))
    put(pv, "is_timestamp_low_eq", int(il == ll))
    put(pv, "inv_timestamp_low", 0 if il == ll else inv(ll - il))
    put(pv, "is_first_execution_shard", int(initial_timestamp == 1))
    return pv


def initialized_state(pc_start_abs):
    """A precompile shard's public values: `update_initialized_state` (public_values.rs:L272-L299) on a fresh record — it is a
    non-execution shard in the program's INITIAL state (timestamp 1, pc = the entry point, nothing committed, exit code 0)."""
    pv = set_state(blank(), pc_start_abs, pc_start_abs, 1, 1, 0, False)
    put(pv, "last_timestamp_inv", 0)
    put(pv, "is_first_execution_shard", 0)
    return pv


def finalized_state(timestamp, pc, exit_code, committed_value_digest, deferred_proofs_digest):
    """A memory shard's public values: `update_finalized_state` (public_values.rs:L301-L345) then `finalize_public_values(false)`
    (worker/prover/core.rs:L370-L397) — a non-execution shard in the program's FINAL state, both commit system calls made."""
    pv = set_state(blank(), pc, pc, timestamp, timestamp, exit_code, False)
    put(pv, "prev_exit_code", exit_code)
    for name, words in (("committed_value_digest", digest_bytes(committed_value_digest)), ("deferred_proofs_digest", list(deferred_proofs_digest))):
        put(pv, "prev_" + name, words)
        put(pv, name, words)
    for name in ("prev_commit_syscall", "commit_syscall", "prev_commit_deferred_syscall", "commit_deferred_syscall"):
        put(pv, name, 1)
    return pv


def set_global(pv, count, cumulative_sum):
    """global_count / global_cumulative_sum: the Global chip's real rows and the digest its last real row accumulates to
    (global/mod.rs:L119, the `SepticDigest::zero()` start when it has none)."""
    put(pv, "global_count", count)
    put(pv, "global_cumulative_sum", cumulative_sum if cumulative_sum is not None else list(R.CURVE_CUMULATIVE_SUM_START[0]) + list(R.CURVE_CUMULATIVE_SUM_START[1]))
    return pv


def no_memory_events(pv, init_addr=0, finalize_addr=0):
    """A shard without MemoryGlobalInit / Finalize rows: both address chains stand still (record.rs split: previous == last)."""
    for name, a in (("previous_init_addr", init_addr), ("last_init_addr", init_addr), ("previous_finalize_addr", finalize_addr), ("last_finalize_addr", finalize_addr)):
        put(pv, name, addr_limbs(a))
    put(pv, "global_init_count", 0)
    put(pv, "global_finalize_count", 0)
    return pv


def to_tensor(pv):
    """What `MachineRecord::public_values` hands the prover: PROOF_MAX_NUM_PVS words, zero beyond the machine's 160."""
    out = torch.zeros(PROOF_MAX_NUM_PVS, dtype=torch.int64)
    out[:NUM_PV_ELTS] = torch.as_tensor([int(v) % P for v in pv], dtype=torch.int64)
    return out


def row(publics, device="cpu"):
    """The one-row table the interaction program reads: [1, 160] canonical int64."""
    t = torch.as_tensor(publics, dtype=torch.int64)[:NUM_PV_ELTS]
    return t.reshape(1, NUM_PV_ELTS).to(device)


# ---------------------------------------------------------------------------------------------------------------------
# the checks across the shards of one proof (crates/prover/src/verify.rs:L106-L520, everything before the per-shard verify_shard)
def _digest_sum(digests):
    """`Sum for SepticDigest` (septic_digest.rs:L95-L110): start + sum (digest_i - offset) + offset - start."""
    from . import septic as SE
    I64 = torch.int64
    pt = lambda xy: (torch.tensor([xy[0]], dtype=I64), torch.tensor([xy[1]], dtype=I64))
    neg = lambda p: (p[0], (P - p[1]) % P)
    zero = pt(R.CURVE_CUMULATIVE_SUM_START)
    start = pt(DIGEST_SUM_START)
    acc = start
    for d in digests:
        acc = SE.ec_add(SE.ec_add(acc, pt((d[:7], d[7:]))), neg(zero))
    acc = SE.ec_add(SE.ec_add(acc, zero), neg(start))
    return [int(v) for v in acc[0][0]] + [int(v) for v in acc[1][0]]


DIGEST_SUM_START = ([0x1732050, 0x8075688, 0x7729352, 0x7446341, 0x5058723, 0x6694280, 0x5253810],
                    [1095433104, 7540207, 1124564165, 2035506693, 11121645, 102781365, 398772161])   # septic_digest.rs:L19-L26


def verify_proof_public_values(shards, vk_pc_start, initial_global_cumulative_sum=None):
    """`SP1Prover::verify` up to the per-shard proofs: `shards` = the public values of a core proof's shards in proof order
    (each PROOF_MAX_NUM_PVS words). Returns None when they chain, else the reference's error string."""
    f = lambda pv, name: get(pv, name)
    if not shards:
        return "empty proof"
    if any(len(pv) != PROOF_MAX_NUM_PVS for pv in shards):
        return "invalid public values length"
    firsts = [f(pv, "is_first_execution_shard") for pv in shards]
    if any(x not in (0, 1) for x in firsts):
        return "is_first_execution_shard is not boolean"
    if sum(firsts) > 1:
        return "is_first_execution_shard is set to one for multiple shards"
    if sum(firsts) == 0:
        return "first execution shard is not set"
    prev_ts = [0, 0, 0, 1]
    for pv in shards:
        if f(pv, "initial_timestamp") != prev_ts:
            return "invalid initial timestamp"
        moved = f(pv, "initial_timestamp") != f(pv, "last_timestamp")
        if f(pv, "is_execution_shard") != 0 and not moved:
            return "timestamp should change on execution shard"
        if f(pv, "is_execution_shard") != 1 and moved:
            return "timestamp should not change on non-execution shard"
        prev_ts = f(pv, "last_timestamp")
    prev_pc = addr_limbs(vk_pc_start)
    for i, pv in enumerate(shards):
        if f(pv, "pc_start") != prev_pc:
            return "pc_start != vk.pc_start: program counter should start at vk.pc_start" if i == 0 else \
                "pc_start != prev_next_pc: pc_start should equal prev_next_pc for all shards"
        if f(pv, "is_execution_shard") != 1 and f(pv, "pc_start") != f(pv, "next_pc"):
            return "pc_start != next_pc: pc_start should equal next_pc for non-execution shards"
        prev_pc = f(pv, "next_pc")
    if prev_pc != [HALT_PC, 0, 0]:
        return "next_pc != HALT_PC: execution should have halted"
    prev_exit = 0
    for pv in shards:
        if f(pv, "prev_exit_code") != prev_exit:
            return "public_values.prev_exit_code != prev_exit_code: prev_exit_code does not match previous shard's exit_code"
        if f(pv, "is_execution_shard") != 1 and f(pv, "prev_exit_code") != f(pv, "exit_code"):
            return "prev_exit_code != exit_code: exit code should be same in non-execution shards"
        if f(pv, "prev_exit_code") != 0 and f(pv, "prev_exit_code") != f(pv, "exit_code"):
            return "prev_exit_code != exit_code: exit code should change at most once"
        prev_exit = f(pv, "exit_code")
    if any(f(pv, "proof_nonce") != f(shards[0], "proof_nonce") for pv in shards[1:]):
        return "proof_nonce != proof_nonce_first_shard"
    last = {n: [0, 0, 0] for n in ("init_addr", "finalize_addr", "init_page_idx", "finalize_page_idx")}
    for pv in shards:
        for n in ("init_addr", "finalize_addr", "init_page_idx", "finalize_page_idx"):
            if f(pv, "previous_" + n) != last[n]:
                return "previous_%s != last_%s_prev" % (n, n)
        if f(pv, "is_untrusted_programs_enabled") != 0:
            return "public_values.is_untrusted_programs_enabled != vk.untrusted_config.enable_untrusted_programs"
        for n in last:
            last[n] = f(pv, "last_" + n)
    if last["init_addr"] == [0, 0, 0]:
        return "the zero address was never initialized"
    if last["finalize_addr"] == [0, 0, 0]:
        return "the zero address was never finalized"
    prev = {"committed_value_digest": [0] * 32, "deferred_proofs_digest": [0] * 8, "commit_syscall": 0, "commit_deferred_syscall": 0}
    for pv in shards:
        for n in prev:
            if f(pv, "prev_" + n) != prev[n]:
                return "prev_%s doesn't equal the previous shard's %s" % (n, n)
        for n in prev:
            prev[n] = f(pv, n)
    if prev["commit_syscall"] != 1:
        return "COMMIT syscall was never called"
    if prev["commit_deferred_syscall"] != 1:
        return "COMMIT_DEFERRED_PROOFS syscall was never called"
    zero = list(R.CURVE_CUMULATIVE_SUM_START[0]) + list(R.CURVE_CUMULATIVE_SUM_START[1])
    digests = [initial_global_cumulative_sum if initial_global_cumulative_sum is not None else zero]
    digests += [f(pv, "global_cumulative_sum") for pv in shards]
    if _digest_sum(digests) != zero:
        return "global cumulative sum is not zero"
    return None


def verifier_program():
    """The machine's eval_public_values as a verifier needs it (the `pv_program` of the restated verify_shard in the tests):
    (constraints, interactions, `max_interaction_kinds_values`, PROOF_MAX_NUM_PVS)."""
    air, it = program()
    return air, it, max_interaction_arity(), PROOF_MAX_NUM_PVS
