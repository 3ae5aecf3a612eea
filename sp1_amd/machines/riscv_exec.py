"""Real guest programs: the rv64im executor of libsp1hip.so (sp1_amd/csrc/rv64_exec.cpp) and, from its events, the tables of
every chip of a core shard — what `MinimalExecutor` + `TracingVM` + the chips' `generate_trace_into` do in the reference
(/root/reference/crates/core/executor/src/{minimal,tracing,vm}.rs; crates/core/machine/src/**/trace.rs as cited in
riscv_trace.py, whose column fillers this module drives with executed events instead of the synthetic loop body).

    ex = Executor(open(elf, "rb").read(), stdin=[n.to_bytes(4, "little")])
    for shard in ex.shards(max_cycles=1 << 21):               # ExecutedShard: events, local memory, public values
        machine, tables, publics = shard_tables(ex, shard, device)   # every chip's (prep, main), ready for api.prove_shard

The proof statement per shard is the reference's: the chips of the shard's shape cluster (those without events at height zero),
their constraints on every row, and the Byte / Range / Program / Memory / State buses balanced against the messages the shard's
PUBLIC VALUES send and receive (`eval_public_values`, public_values.py: the initial and final CPU state, the ends of the Global
accumulation and of the memory-initialisation chains); what crosses shards (memory states, syscalls) goes through the Global chip's
digest, and the public values chain from shard to shard as `SP1Prover::verify` requires (public_values.verify_proof_public_values).
"""
import ctypes as C
import gzip
import os
import types

import numpy as np
import torch

from .. import _lib
from . import public_values as PVM
from . import riscv as R
from . import riscv_trace as RT
from .riscv_trace import I64, MASK16, OPC, P, POS_OFF, Table, limbs16

EV_WORDS, KECCAK_WORDS, POSEIDON2_WORDS, SHA_EXTEND_WORDS, SHA_COMPRESS_WORDS, UINT256_WORDS = 20, 77, 26, 786, 155, 31
SECP_ADD_WORDS, SECP_DOUBLE_WORDS = 43, 26
# kind: (SP1HIP_RV64_FAMILY_*, chip) — the field / curve precompiles behind sp1hip_rv64_precompile_events (include/sp1hip.h)
FAMILIES = {"secp256r1_add": (0, "Secp256r1AddAssign"), "secp256r1_double": (1, "Secp256r1DoubleAssign"), "bn254_add": (2, "Bn254AddAssign"),
            "bn254_double": (3, "Bn254DoubleAssign"), "bls12381_add": (4, "Bls12381AddAssign"), "bls12381_double": (5, "Bls12381DoubleAssign"),
            "bn254_fp": (6, "Bn254FpOpAssign"), "bls12381_fp": (7, "Bls12381FpOpAssign"), "bn254_fp2_addsub": (8, "Bn254Fp2AddSubAssign"),
            "bls12381_fp2_addsub": (9, "Bls12381Fp2AddSubAssign"), "bn254_fp2_mul": (10, "Bn254Fp2MulAssign"), "bls12381_fp2_mul": (11, "Bls12381Fp2MulAssign"),
            "ed_add": (12, "EdAddAssign"), "ed_decompress": (13, "EdDecompress"), "uint256_ops": (14, "Uint256Ops")}
(E_PC, E_CLK, E_OP, E_OPA, E_OPB, E_OPC, E_FLAGS, E_A, E_B, E_C, E_A_PREV, E_A_PTS, E_B_PTS, E_C_PTS, E_MADDR, E_M_PTS, E_M_PREV, E_M_NEW,
 E_NEXT_PC, E_SPARE) = range(EV_WORDS)


GUESTS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench", "programs")


def guest_file(name):
    """The bytes of bench/programs/<name>.gz: the reference's guest binaries (and rsp's input) are stored gzip-compressed,
    unmodified otherwise (bench/programs/README.md)."""
    with gzip.open(os.path.join(GUESTS_DIR, name + ".gz"), "rb") as f:
        return f.read()


class ExecutedShard:
    """One shard of an execution: `events` [n, 20] / `local` [m, 5] / `keccak` [k, 77] int64 arrays (two's complement images of
    the executor's u64 words) and the shard's public-value fields."""

    def __init__(self, info, events, local, keccak, poseidon2, sha_extend, sha_compress, uint256, secp_add, secp_double):
        self.index, self.cycles = int(info.shard), int(info.n_cycles)
        self.events, self.local, self.keccak, self.poseidon2 = events, local, keccak, poseidon2
        self.sha_extend, self.sha_compress, self.uint256 = sha_extend, sha_compress, uint256
        self.secp256k1_add, self.secp256k1_double = secp_add, secp_double
        self.families = {}                                        # kind (FAMILIES) -> [n, words] events, only the kinds that occurred
        self.pc_start, self.next_pc = int(info.pc_start), int(info.next_pc)
        self.clk_start, self.clk_end = int(info.clk_start), int(info.clk_end)
        self.halted, self.exit_code = bool(info.halted), int(info.exit_code)
        self.commit_syscall, self.commit_deferred_syscall = int(info.commit_syscall), int(info.commit_deferred_syscall)
        self.committed_value_digest = [int(x) for x in info.committed_value_digest]
        self.deferred_proofs_digest = [int(x) for x in info.deferred_proofs_digest]
        self.estimated_area, self.estimated_max_height = int(info.estimated_area), int(info.estimated_max_height)


class Executor:
    def __init__(self, elf, stdin=()):
        self.lib = _lib.load()
        buf = (C.c_uint8 * len(elf)).from_buffer_copy(elf)
        h = C.c_void_p()
        _lib.check(self.lib.sp1hip_rv64_create(buf, len(elf), C.byref(h)))
        self.h = h
        for entry in stdin:
            self.write_stdin(entry)
        self.halted = False

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.sp1hip_rv64_destroy(self.h)
            self.h = None

    def write_stdin(self, data):
        data = bytes(data)
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
        _lib.check(self.lib.sp1hip_rv64_write_stdin(self.h, buf, len(data)))

    @staticmethod
    def _matrix(ptr, rows, cols, copy=True):
        if rows == 0:
            return np.zeros((0, cols), dtype=np.int64)
        m = np.ctypeslib.as_array(ptr, shape=(rows * cols,)).view(np.int64).reshape(rows, cols)
        return m.copy() if copy else m

    def run_shard(self, max_cycles, record=True, copy=True):
        """The next shard. record=False: run it without keeping its instruction events (`events` comes back empty).
        copy=False: `events` is a view of the executor's buffer, valid until the next run_shard (a full shard is 1.3 GB)."""
        _lib.check(self.lib.sp1hip_rv64_set_recording(self.h, int(bool(record))))
        info = _lib.Rv64ShardInfo()
        _lib.check(self.lib.sp1hip_rv64_run_shard(self.h, int(max_cycles), C.byref(info)))
        n_events = info.n_events if record else 0
        shard = ExecutedShard(info, self._matrix(self.lib.sp1hip_rv64_events(self.h), n_events, EV_WORDS, copy),
                              self._matrix(self.lib.sp1hip_rv64_local_memory(self.h), info.n_local, 5),
                              self._matrix(self.lib.sp1hip_rv64_keccak_events(self.h), info.n_keccak, KECCAK_WORDS),
                              self._matrix(self.lib.sp1hip_rv64_poseidon2_events(self.h), info.n_poseidon2, POSEIDON2_WORDS),
                              self._matrix(self.lib.sp1hip_rv64_sha_extend_events(self.h), info.n_sha_extend, SHA_EXTEND_WORDS),
                              self._matrix(self.lib.sp1hip_rv64_sha_compress_events(self.h), info.n_sha_compress, SHA_COMPRESS_WORDS),
                              self._matrix(self.lib.sp1hip_rv64_uint256_events(self.h), info.n_uint256, UINT256_WORDS),
                              self._matrix(self.lib.sp1hip_rv64_secp256k1_add_events(self.h), info.n_secp256k1_add, SECP_ADD_WORDS),
                              self._matrix(self.lib.sp1hip_rv64_secp256k1_double_events(self.h), info.n_secp256k1_double, SECP_DOUBLE_WORDS))
        n_ev, n_words, data = C.c_uint64(), C.c_uint64(), C.POINTER(C.c_uint64)()
        for kind, (family, _) in FAMILIES.items():
            _lib.check(self.lib.sp1hip_rv64_precompile_events(self.h, family, C.byref(n_ev), C.byref(n_words), C.byref(data)))
            if n_ev.value:
                shard.families[kind] = self._matrix(data, n_ev.value, n_words.value)
        self.halted = shard.halted
        return shard

    def cut_by_area(self, element_threshold=None, height_threshold=None):
        """Shards end where the reference's executor would end them (`ShapeChecker`, vm/shapes.rs; thresholds opts.rs:L12-L14):
        when the estimated trace area or a table's height reaches its threshold. The cost model — columns per chip — is this
        package's chips' (pinned to rv64im_costs.json); the estimator itself runs inside the executor, instruction by instruction."""
        cost = lambda name: (lambda a: a.main_width + a.prep_width)(R.chip(name)[0])
        lim = _lib.Rv64ShardLimits()
        lim.element_threshold = ELEMENT_THRESHOLD if element_threshold is None else element_threshold
        lim.height_threshold = HEIGHT_THRESHOLD if height_threshold is None else height_threshold
        rows = self.program()[1].shape[0]
        lim.fixed_area = -(-rows // 32) * 32 * cost("Program") + (1 << 16) * cost("Byte") + (1 << 17) * cost("Range")
        chips = {}
        for name, o in OPC.items():
            if o <= OPC["REMUW"]:
                chip = _ALU_CHIP[o]
            elif name in _MEM_CHIP:
                chip = _MEM_CHIP[name]
            elif name in RT.BRANCH_OPS:
                chip = "Branch"
            else:
                chip = {"JAL": "Jal", "JALR": "Jalr", "AUIPC": "UType", "LUI": "UType", "ECALL": "SyscallInstrs"}.get(name)
            if chip is not None:
                lim.opcode_cost[o], lim.opcode_chip[o] = cost(chip), chips.setdefault(chip, len(chips))
        lim.alu_x0_cost, lim.load_x0_cost, lim.memory_local_cost, lim.global_cost = cost("AluX0"), cost("LoadX0"), cost("MemoryLocal"), cost("Global")
        lim.syscall_core_cost, lim.memory_bump_cost, lim.state_bump_cost = cost("SyscallCore"), cost("MemoryBump"), cost("StateBump")
        _lib.check(self.lib.sp1hip_rv64_set_shard_limits(self.h, C.byref(lim)))
        return self

    def shards(self, max_cycles):
        while not self.halted:
            yield self.run_shard(max_cycles)

    def program(self):
        """(pc_base, [n, 6] int64: opcode, op_a, op_b, op_c, imm_b, imm_c)."""
        base, n, tab = C.c_uint64(), C.c_uint64(), C.POINTER(C.c_uint64)()
        _lib.check(self.lib.sp1hip_rv64_program(self.h, C.byref(base), C.byref(n), C.byref(tab)))
        return int(base.value), self._matrix(tab, n.value, 6)

    def global_memory(self):
        """[t, 4] int64 per address the run touched: address, initial value, final value, final timestamp."""
        n, tab = C.c_uint64(), C.POINTER(C.c_uint64)()
        _lib.check(self.lib.sp1hip_rv64_global_memory(self.h, C.byref(n), C.byref(tab)))
        return self._matrix(tab, n.value, 4)

    def memory_image(self):
        """[m, 2] int64: (address, value) of every 8-byte word of the ELF's segments, ascending (`Program::memory_image`)."""
        n, tab = C.c_uint64(), C.POINTER(C.c_uint64)()
        _lib.check(self.lib.sp1hip_rv64_memory_image(self.h, C.byref(n), C.byref(tab)))
        return self._matrix(tab, n.value, 2)

    def output(self, which=0):
        p, n = _lib.u8p(), C.c_uint64()
        _lib.check(self.lib.sp1hip_rv64_output(self.h, which, C.byref(p), C.byref(n)))
        return bytes(np.ctypeslib.as_array(p, shape=(n.value,))) if n.value else b""


# ---------------------------------------------------------------------------------------------------------------------
# events -> tables
_OPN = {v: k for k, v in OPC.items()}
_ALU_CHIP = {}
for _chip, _ops in RT.ALU_KINDS.items():
    for _o in _ops:
        _ALU_CHIP[OPC[_o]] = _chip
_MEM_CHIP = {"LB": "LoadByte", "LBU": "LoadByte", "LH": "LoadHalf", "LHU": "LoadHalf", "LW": "LoadWord", "LWU": "LoadWord", "LD": "LoadDouble",
             "SB": "StoreByte", "SH": "StoreHalf", "SW": "StoreWord", "SD": "StoreDouble"}
PV, PV_WORDS = PVM.PV, PVM.NUM_PV_ELTS                # PublicValues word offsets (public_values.py)


def chip_of_events(ev):
    """The table every executed instruction goes to (tracing.rs: emit_alu_event L1153-L1228, emit_mem_instr_event L1096-L1150)."""
    op, rd = ev[:, E_OP], ev[:, E_OPA]
    names = np.empty(len(op), dtype=object)
    for o in np.unique(op):
        nm = _OPN[int(o)]
        m = op == o
        if o <= OPC["REMUW"]:
            names[m & (rd != 0)] = _ALU_CHIP[int(o)]
            names[m & (rd == 0)] = "AluX0"
        elif nm in _MEM_CHIP:
            names[m] = _MEM_CHIP[nm]
            if nm.startswith("L"):
                names[m & (rd == 0)] = "LoadX0"
        elif nm in RT.BRANCH_OPS:
            names[m] = "Branch"
        else:
            names[m] = {"JAL": "Jal", "JALR": "Jalr", "AUIPC": "UType", "LUI": "UType", "ECALL": "SyscallInstrs"}[nm]
    return names


class EventView:
    """An `ExecutedShard` behind the interface riscv_trace.Tracer's column fillers read an execution through: one "iteration"
    (K = 1) whose positions are the shard's executed instructions."""
    K = 1

    def __init__(self, shard, pc_base, device):
        self.dev = torch.device(device)
        ev = torch.as_tensor(shard.events, device=self.dev)
        self.ev, self.L, self.clk0 = ev, ev.shape[0], shard.clk_start
        self.op = ev[:, E_OP]
        imm_b, imm_c = (ev[:, E_FLAGS] & 1) == 1, (ev[:, E_FLAGS] & 2) == 2
        self.has_imm = imm_c
        self.imm = torch.where(self.op == OPC["JAL"], ev[:, E_OPB], ev[:, E_OPC])
        neg = torch.full_like(self.op, -1)
        self.slot_reg = {"A": ev[:, E_OPA], "B": torch.where(imm_b, neg, ev[:, E_OPB]), "C": torch.where(imm_c, neg, ev[:, E_OPC])}
        self.W = ev[:, E_A][None, :]
        self.values = {"A": ev[:, E_A_PREV], "B": ev[:, E_B], "C": ev[:, E_C]}
        self.prev_ts = {"A": ev[:, E_A_PTS], "B": ev[:, E_B_PTS], "C": ev[:, E_C_PTS]}
        self.clk_inc = torch.where(self.op == OPC["ECALL"], 8 + RT.ECALL_EXTRA_CLK, 8)
        self.body = types.SimpleNamespace(chip=chip_of_events(shard.events), pc_base=pc_base)

    def T(self, k, p):
        return self.ev[p, E_CLK]

    def pc(self, p):
        return self.ev[p, E_PC]

    def reg_value(self, k, p, slot):
        return self.values[slot][p]

    def _grid(self, positions):
        p = torch.as_tensor(np.asarray(positions, dtype=np.int64), device=self.dev)
        return torch.zeros_like(p), p


class EventTracer(RT.Tracer):
    """riscv_trace.Tracer over executed events. The per-chip column fillers (adapters, ALU / shift / multiply / divide
    operations, load / store chips, SyscallInstrs, Global) are the base class's; what changes is where the timeline comes from:
    previous timestamps and memory states are the executor's records instead of the loop body's static analysis."""

    def __init__(self, executor, shard, device="cpu", prev=None):
        """prev: the public values of the previous shard in execution order (a list of words) or None for the first — where
        prev_committed_value_digest, prev_exit_code and the prev_commit_* flags come from (verify.rs:L262-L296, L445-L480)."""
        self.pc_base, self.program = executor.program()
        self.shard, self.prev = shard, prev
        super().__init__(EventView(shard, self.pc_base, device))
        self.real_global = True

    def _prev_ts(self, k, p, slot):
        return self.ex.prev_ts[slot][p]

    def build(self):
        def alu_x0(tb, k, p, a, bv, cv):
            tb.set("opcode", self.ex.op[p])
            tb.set("is_real", 1)
        self.simple_alu("AluX0", "ALU", alu_x0)
        return super().build()

    def branch(self):
        super().branch()
        if "Branch" in self.tables:
            _, p = self.rows_of("Branch")
            self.tables["Branch"].set("next_pc", limbs16(self.ex.ev[p, E_NEXT_PC])[:, :3])

    def jalr(self):
        super().jalr()
        if "Jalr" in self.tables:
            _, p = self.rows_of("Jalr")
            tb, rd0 = self.tables["Jalr"], self.ex.slot_reg["A"][p] == 0
            tb.set("op_a_value", limbs16(self.ex.pc(p) + 4) * (~rd0).to(I64)[:, None])

    def memory_instructions(self):
        ex, ev = self.ex, self.ex.ev
        self.mem_words = None
        for c in list(RT.LOAD_KINDS) + list(RT.STORE_KINDS):
            k, p = self.rows_of(c)
            if len(p):
                self.fill_mem_chip(c, k, p, ev[p, E_MADDR], ev[p, E_CLK] + POS_OFF["M"], ev[p, E_M_PTS], ev[p, E_M_PREV], ev[p, E_M_NEW],
                                   ex.values["A"][p])

    def state_chain(self):
        ex, ev, dev = self.ex, self.ex.ev, self.dev
        T, pc, nxt_pc, inc = ev[:, E_CLK], ev[:, E_PC], ev[:, E_NEXT_PC], ex.clk_inc
        op = ex.op
        halt = (op == OPC["ECALL"]) & ((ev[:, E_A_PREV] & 0xFF) == 0)
        normal_pc = ((op >= OPC["BEQ"]) & (op <= OPC["JALR"])) | halt                    # these rows send next_pc in normal form
        pc_carry = (~normal_pc) & (((pc & MASK16) + 4) > MASK16)
        clk_carry = ((T & 0xFFFFFF) + inc) >= (1 << 24)
        need = pc_carry | clk_carry
        self.final_state = (self.shard.clk_end, self.shard.next_pc)
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
            sent[:, 0] += 4
            tb.set("pc", torch.where(pcc[:, None], sent, limbs16(nxt)[:, :3]))
            tb.set("is_clk", ((nT >> 24) != (Tn >> 24)).to(I64))
            tb.set("is_real", 1)

    def memory_local_and_bumps(self):
        dev = self.dev
        loc = torch.as_tensor(self.shard.local, device=dev)
        addr, it, iv, ft, fv = (loc[:, i] for i in range(5))
        air, _ = R.chip("MemoryLocal")
        tb = Table(air, len(addr), dev)
        self.tables["MemoryLocal"] = tb
        tb.set("addr", limbs16(addr)[:, :3])
        tb.set("initial_clk_high", it >> 24)
        tb.set("initial_clk_low", it & 0xFFFFFF)
        tb.set("final_clk_high", ft >> 24)
        tb.set("final_clk_low", ft & 0xFFFFFF)
        for tag, v in (("initial", iv), ("final", fv)):
            l = limbs16(v)
            tb.set(tag + "_value", l)
            tb.set(tag + "_value_lower", l[:, 2] & 0xFF)
            tb.set(tag + "_value_upper", l[:, 2] >> 8)
        tb.set("is_real", 1)
        self._bump_rows()

    def _program_table(self):
        return RT.program_table(self.program, self.pc_base, self.dev, executed_pc=self.ex.ev[:, E_PC])

    def public_values(self):
        return execution_public_values(self.shard, self.prev)

    def finish(self):
        ex, dev, sh = self.ex, self.dev, self.shard
        machine = {name: R.chip(name) for name in self.tables}
        pv = self.public_values()
        ml = self.tables["MemoryLocal"]
        (_, recv, _), (_, send, _) = RT.eval_interactions(R.chip("MemoryLocal")[1], ml.main[:ml.n], None, kinds=(R.GLOBAL,))
        events = [torch.stack([recv, send], dim=1).reshape(-1, 11)]
        if "SyscallCore" in self.tables:
            t_ = self.tables["SyscallCore"]
            events += [v for _, v, _ in RT.eval_interactions(R.chip("SyscallCore")[1], t_.main[:t_.n], None, kinds=(R.GLOBAL,))]
        self.global_events = torch.cat(events)
        PVM.set_global(pv, *self.global_chip(machine, self.global_events))
        self.tables["Program"], machine["Program"] = self._program_table()
        self.byte_range_tables(machine, pv)
        self.fill_cluster(machine, RT.smallest_cluster(machine))
        self.pv = pv
        names = sorted(machine)
        return [machine[n] for n in names], {n: (self.tables[n].prep, self.tables[n].main) for n in names}, PVM.to_tensor(pv)


ALU_TRACEGEN_CHIPS = ("Add", "Addi", "Sub", "Addw", "Subw", "Mul", "ShiftRight", "Branch")   # api.RISCV_ALU_CHIPS: tables made on the device


def pack_alu_events(events, chip=None):
    """The executor's 20-word instruction events -> `sp1hip_rv64_alu_event_t` records (include/sp1hip.h: 11 u64 words — pc, clk,
    ops, a, b, c, a_prev, a_pts, b_pts, c_pts, aux) for the chips whose tables the device generates (api.tracegen_riscv_alu), in
    the events' order = the tables' row order. `events`: [n, 20] int64 (numpy or torch; all of a shard's events when `chip` names
    the chip to select, else already that chip's). An event is 88 bytes where the row it becomes is 120 (Addi) to 328 (Mul)."""
    ev = events if chip is None else events[np.nonzero(chip_of_events(np.asarray(events.cpu() if torch.is_tensor(events) else events)) == chip)[0]]
    xp = torch if torch.is_tensor(ev) else np
    flags = ev[:, E_FLAGS]
    imm_c = (flags & 2) >> 1
    ops = ev[:, E_OP] | (ev[:, E_OPA] << 8) | ((ev[:, E_OPB] & 0xFF) << 16) | (((1 - imm_c) * (ev[:, E_OPC] & 0xFF)) << 24) | ((flags & 3) << 32)
    c = xp.where(imm_c == 1, ev[:, E_OPC], ev[:, E_C])                   # an immediate operand travels as the value
    cols = [ev[:, E_PC], ev[:, E_CLK], ops, ev[:, E_A], ev[:, E_B], c, ev[:, E_A_PREV], ev[:, E_A_PTS], ev[:, E_B_PTS], ev[:, E_C_PTS], ev[:, E_NEXT_PC]]
    return xp.stack(cols, 1) if xp is np else torch.stack(cols, dim=1).contiguous()


def execution_public_values(sh, prev=None):
    """An execution shard's `PublicValues` as the tracing executor leaves them (tracing.rs postprocess L548-L562, executor.rs:L60
    finalize_public_values(true)) with the previous shard's state threaded in (`prev`: its words, None for the first shard); the
    Global chip's two fields are set once its table exists (EventTracer.finish)."""
    pv = PVM.set_state(PVM.blank(), sh.pc_start, sh.next_pc, sh.clk_start, sh.clk_end, sh.exit_code, True)
    PVM.put(pv, "committed_value_digest", PVM.digest_bytes(sh.committed_value_digest))
    PVM.put(pv, "deferred_proofs_digest", sh.deferred_proofs_digest)
    PVM.put(pv, "commit_syscall", sh.commit_syscall)
    PVM.put(pv, "commit_deferred_syscall", sh.commit_deferred_syscall)
    if prev is not None:
        for name in ("committed_value_digest", "deferred_proofs_digest", "exit_code", "commit_syscall", "commit_deferred_syscall"):
            PVM.put(pv, "prev_" + name, PVM.get(prev, name))
    return PVM.no_memory_events(pv)


def shard_tables(executor, shard, device="cpu", prev=None):
    """(machine, tables, public values) of one executed shard: machine = [(AirProgram, InteractionProgram)] in chip-name order —
    the shard's whole shape cluster —, tables = {name: (prep, main)} canonical int64 tensors on `device`, public values = the
    PROOF_MAX_NUM_PVS words of its ShardProof (`prev`: the previous shard's words, see EventTracer)."""
    return EventTracer(executor, shard, device, prev=prev).build()


def proof_order(kinds):
    """The order the shards of a run stand in inside a core proof, as indices into `kinds` (what `program_shards` yielded): the
    precompile shards first — they are in the program's INITIAL state (timestamp 1, pc = entry) —, the core shards in execution
    order, then the memory shards in the FINAL state (crates/prover/src/verify.rs:L160-L260 only accepts that chain)."""
    idx = range(len(kinds))
    return [i for i in idx if kinds[i] not in ("core", "memory")] + [i for i in idx if kinds[i] == "core"] + [i for i in idx if kinds[i] == "memory"]


ELEMENT_THRESHOLD, HEIGHT_THRESHOLD = (1 << 28) + (1 << 27), 1 << 22        # core/executor/src/opts.rs:L12-L14
# kind: (chip, control chip | None, rows per event, touched addresses) — air.rs:L473-L480, syscall_code.rs:L439-L479
PRECOMPILES = {"keccak": ("KeccakPermute", "KeccakPermuteControl", 24, 25), "poseidon2": ("Poseidon2", None, 1, 8),
               "sha_extend": ("ShaExtend", "ShaExtendControl", 48, 64), "sha_compress": ("ShaCompress", "ShaCompressControl", 80, 72),
               "uint256": ("Uint256MulMod", None, 1, 12), "secp256k1_add": ("Secp256k1AddAssign", None, 1, 16),
               "secp256k1_double": ("Secp256k1DoubleAssign", None, 1, 8),
               **{k: (FAMILIES[k][1], None, 1, t) for k, t in (("secp256r1_add", 16), ("secp256r1_double", 8), ("bn254_add", 16), ("bn254_double", 8),
                   ("bls12381_add", 24), ("bls12381_double", 12), ("bn254_fp", 8), ("bls12381_fp", 12), ("bn254_fp2_addsub", 16), ("bls12381_fp2_addsub", 24),
                   ("bn254_fp2_mul", 16), ("bls12381_fp2_mul", 24), ("ed_add", 16), ("ed_decompress", 8), ("uint256_ops", 20))}}


def split_thresholds(program_rows):
    """`SplitOpts::new` (core/executor/src/opts.rs:L186-L240): how many events of one system call, and how many memory
    initialise / finalise events, go into one shard — the trace area left beside the fixed tables divided by the cost of one event
    (its chip rows, control row, MemoryLocal / Global rows per touched address, SyscallPrecompile row: utils.rs:L117-L154),
    bounded by the height limit, rounded down to 32."""
    cost = lambda name: (lambda a: a.main_width + a.prep_width)(R.chip(name)[0])
    trunc = lambda v: v // 32 * 32
    area = ELEMENT_THRESHOLD - (-(-program_rows // 32) * 32 * cost("Program") + (1 << 16) * cost("Byte") + (1 << 17) * cost("Range"))
    out = {}
    for kind, (chip_name, control, rows, touched) in PRECOMPILES.items():
        per = rows * cost(chip_name) + (cost(control) if control else 0) + touched * cost("MemoryLocal") + 2 * touched * cost("Global")
        per += cost("SyscallPrecompile") + cost("Global")
        out[kind] = min(trunc(area // per), trunc(HEIGHT_THRESHOLD // max(rows, 2 * touched + 1)))
    out["memory"] = trunc(min(area // (cost("MemoryGlobalInit") + cost("Global")), HEIGHT_THRESHOLD) // 2)
    return out


def program_shards(executor, max_cycles, device="cpu", core_limit=None):
    """Every shard of a run, in the order the reference's controller emits them: the core shards as the program executes
    (`(kind, machine, tables, publics, global events, ExecutedShard)` with kind = "core"), then one precompile shard for the
    KECCAK_PERMUTE, POSEIDON2, SHA_EXTEND and SHA_COMPRESS calls each if there were any ("keccak", "poseidon2", "sha_extend",
    "sha_compress"), then the memory shard: MemoryGlobalInit / MemoryGlobalFinalize over
    every address the run touched ("memory"). The global events of all shards cancel as a multiset: that is the statement the
    shards' septic-curve digests add up to. `core_limit`: only the first so many core shards are traced and yielded (the rest of
    the program still runs, so every precompile and memory shard is there)."""
    from . import riscv_more_trace as MT
    keccak, poseidon2, sha_extend, sha_compress, uint256, secp_add, secp_double = [], [], [], [], [], [], []
    families = {}
    n_core = 0
    prev_pv, ctx, shard = None, None, None
    while not executor.halted:
        keep = core_limit is None or n_core < core_limit     # beyond the limit: executed (their precompile calls count), not traced
        shard = executor.run_shard(max_cycles, record=keep)
        n_core += 1
        if ctx is None:                                      # what the shards without instructions take from the run (RunContext)
            pc_base, program = executor.program()
            ctx = RT.RunContext(program, pc_base, pc_start=shard.pc_start)
        if keep:
            tr = EventTracer(executor, shard, device, prev=prev_pv)
            machine, tables, publics = tr.build()
            prev_pv = tr.pv
        else:
            prev_pv = execution_public_values(shard, prev_pv)
        if shard.keccak.shape[0]:
            keccak.append(shard.keccak)
        if shard.poseidon2.shape[0]:
            poseidon2.append(shard.poseidon2)
        if shard.sha_extend.shape[0]:
            sha_extend.append(shard.sha_extend)
        if shard.sha_compress.shape[0]:
            sha_compress.append(shard.sha_compress)
        if shard.uint256.shape[0]:
            uint256.append(shard.uint256)
        if shard.secp256k1_add.shape[0]:
            secp_add.append(shard.secp256k1_add)
        if shard.secp256k1_double.shape[0]:
            secp_double.append(shard.secp256k1_double)
        for kind, evs in shard.families.items():
            families.setdefault(kind, []).append(evs)
        if keep:
            yield "core", machine, tables, publics, tr.global_events, shard
            del tr, machine, tables
    limit = split_thresholds(executor.program()[1].shape[0])
    chunks = lambda kind, evs: (lambda a: [a[i:i + limit[kind]] for i in range(0, a.shape[0], limit[kind])])(np.concatenate(evs)) if evs else []
    for kk in chunks("keccak", keccak):
        kk = torch.as_tensor(kk, device=device)
        rd = kk[:, 2:52].reshape(-1, 25, 2)
        machine, tables, publics, gev = MT.precompile_shard_from(kk[:, 0], kk[:, 1], rd[:, :, 1].contiguous(), rd[:, :, 0].contiguous(), device, ctx=ctx)
        yield "keccak", machine, tables, publics, gev, None
    for pp in chunks("poseidon2", poseidon2):
        pp = torch.as_tensor(pp, device=device)
        rd = pp[:, 2:18].reshape(-1, 8, 2)
        machine, tables, publics, gev = MT.poseidon2_shard_from(pp[:, 0], pp[:, 1], rd[:, :, 1].contiguous(), rd[:, :, 0].contiguous(),
                                                               pp[:, 18:26].contiguous(), device, ctx=ctx)
        yield "poseidon2", machine, tables, publics, gev, None
    for name, evs, build in (("sha_extend", sha_extend, MT.sha_extend_shard_from), ("sha_compress", sha_compress, MT.sha_compress_shard_from),
                             ("uint256", uint256, MT.uint256_shard_from), ("secp256k1_add", secp_add, MT.secp256k1_add_shard_from),
                             ("secp256k1_double", secp_double, MT.secp256k1_double_shard_from)):
        for part in chunks(name, evs):
            machine, tables, publics, gev = build(part, device, ctx=ctx)
            yield name, machine, tables, publics, gev, None
    for kind in FAMILIES:                                    # one chip per kind; Fp / UINT256 kinds hold all their system calls' events
        for part in chunks(kind, families.get(kind)):
            machine, tables, publics, gev = MT.family_shard_from(kind, part, device, ctx=ctx)
            yield kind, machine, tables, publics, gev, None
    # ---- the memory shards (prover/src/worker/controller/global.rs:L145-L300). Every touched address — registers with a timestamp,
    # memory words, hinted words — is finalised, and so is every word of the program's memory image, touched or not; every touched
    # address OUTSIDE the image is initialised (registers and fresh memory with 0, hinted words with their hint). The image's own
    # initialisation is the verifying key's `initial_global_cumulative_sum` (image_events / verifying_key_words below). The two
    # streams are merged by address and a shard is flushed when either buffer reaches the threshold: the finalise buffer, which
    # grows at every step, fills first — chunks of `memory` finalise events with the initialise events of their address range.
    gm = executor.global_memory()
    gm = gm[np.argsort(gm[:, 0].astype(np.uint64))]
    if gm.shape[0] == 0 or gm[0, 0] != 0:                  # register x0 opens the address chain whether or not the program read it
        gm = np.concatenate([np.zeros((1, 4), dtype=np.int64), gm])
    img = executor.memory_image()
    t_addr, i_addr = gm[:, 0].astype(np.uint64), img[:, 0].astype(np.uint64)
    outside = ~np.isin(t_addr, i_addr)
    init_addr, init_rec = t_addr[outside], np.stack([gm[outside, 1], np.zeros(int(outside.sum()), dtype=np.int64)], axis=1)
    untouched = ~np.isin(i_addr, t_addr)
    fin_addr = np.concatenate([t_addr, i_addr[untouched]])
    fin_rec = np.concatenate([gm[:, 2:4], np.stack([img[untouched, 1], np.zeros(int(untouched.sum()), dtype=np.int64)], axis=1)])
    order = np.argsort(fin_addr, kind="stable")
    fin_addr, fin_rec = fin_addr[order], fin_rec[order]
    previous_init = previous_fin = 0
    # the memory shards stand in the program's final state (update_finalized_state, worker/prover/core.rs:L370-L397)
    ctx.final = (shard.clk_end, shard.next_pc, shard.exit_code, shard.committed_value_digest, shard.deferred_proofs_digest)
    for at in range(0, fin_addr.shape[0], limit["memory"]):
        fa, fr = fin_addr[at:at + limit["memory"]], fin_rec[at:at + limit["memory"]]
        lo, hi = np.searchsorted(init_addr, fa[0], side="left"), np.searchsorted(init_addr, fa[-1], side="right")
        machine, tables, publics, gev = MT.memory_shard_from(init_addr[lo:hi], init_rec[lo:hi], fr, device, previous_addr=previous_init, ctx=ctx,
                                                             fin_addrs=fa, previous_fin_addr=previous_fin)
        previous_init = int(init_addr[hi - 1]) if hi > lo else previous_init
        previous_fin = int(fa[-1])
        yield "memory", machine, tables, publics, gev, None


def image_events(executor, device="cpu"):
    """The Global events the verifying key stands for: one SEND per word of the program's memory image, in the form MemoryGlobalInit
    would have sent it — [0, 0, address limbs, value limbs 0 / 1 with bytes 4 / 5 on top, value limb 3] at timestamp 0
    (`initial_global_cumulative_sum`, core/executor/src/program.rs:L170-L199). [m, 11] like every shard's events: with them
    appended, the events of a whole run cancel."""
    img = executor.memory_image()
    a, v = torch.as_tensor(img[:, 0], device=device), torch.as_tensor(img[:, 1], device=device)
    lim = lambda x, k: (x >> (16 * k)) & 0xFFFF
    cols = [torch.zeros_like(a), torch.zeros_like(a), lim(a, 0), lim(a, 1), lim(a, 2), lim(v, 0) + (((v >> 32) & 0xFF) << 16), lim(v, 1) + (((v >> 40) & 0xFF) << 16),
            lim(v, 3), torch.ones_like(a), torch.zeros_like(a), torch.full_like(a, R.MEMORY)]
    return torch.stack(cols, dim=1)


def verifying_key_words(executor, pc_start, device="cpu"):
    """What the transcript absorbs of the program besides the preprocessed commitment (`MachineVerifyingKey`: pc_start,
    initial_global_cumulative_sum): the entry point's three 16-bit limbs, then the septic digest x[7], y[7] of the memory image's
    initialisation — canonical integers. pc_start: the first shard's (the ELF's entry point)."""
    from . import septic as SE
    ev = image_events(executor, device)
    start = tuple(torch.tensor(v, dtype=torch.int64, device=device) for v in R.CURVE_CUMULATIVE_SUM_START)
    if ev.shape[0]:
        x, y, _, _ = SE.lift_x(ev[:, :8], ev[:, 10], ev[:, 9] == 1)
        cx, cy = SE.prefix_sums(start, x, y)
        digest = [int(t) for t in cx[-1]] + [int(t) for t in cy[-1]]
    else:
        digest = [int(t) for t in start[0]] + [int(t) for t in start[1]]
    return PVM.addr_limbs(pc_start) + digest


def global_events_balance(event_lists):
    """The cross-shard statement: over all shards, every Global message is sent exactly as often as it is received
    (events [n, 11] = message[8], is_send, is_receive, kind). Returns the messages that do not cancel."""
    ev = torch.cat([e.cpu() for e in event_lists]).numpy()
    if ev.shape[0] > (1 << 18):
        # a large run (millions of messages): first two independent 64-bit multiset fingerprints — sum of (send - receive) *
        # mix(message, kind) with wrap-around —, which are both zero when everything cancels; the exact tally below only otherwise
        u = ev.astype(np.uint64)
        sign = (ev[:, 8] - ev[:, 9]).astype(np.uint64)
        clean = True
        for seed in (0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F):
            h = np.full(ev.shape[0], seed, dtype=np.uint64)
            for col in list(range(8)) + [10]:
                h = (h ^ u[:, col]) * np.uint64(0x100000001B3 | (seed & 0xFFFF0000))
                h ^= h >> np.uint64(29)
            clean &= int((h * sign).sum(dtype=np.uint64)) == 0
        if clean:
            return []
    tally = {}
    for row in ev:
        key = tuple(int(x) for x in row[:8]) + (int(row[10]),)
        tally[key] = tally.get(key, 0) + int(row[8]) - int(row[9])
    return {k: v for k, v in tally.items() if v}
