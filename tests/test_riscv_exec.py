"""Real guest programs (CPU): the rv64im executor of libsp1hip.so (host code: no GPU needed) and the tables made from its events.

* the reference's own guest binaries (bench/programs/*.elf.gz, gzip-compressed but otherwise unmodified copies of /root/reference/sp1-gpu/crates/prover_components/
  programs/*/riscv64im-succinct-zkvm-elf — workload inputs, see bench/programs/README.md) run to HALT with exit code 0; the
  public-values digest the guest COMMITs (SHA-256 computed by some thousands of executed rv64im instructions) equals hashlib's
  over the bytes it wrote to the public-values descriptor;
* on every shard of those runs — core shards, the KECCAK_PERMUTE precompile shard, the memory shard — every constraint of every
  chip vanishes on every row and every bus balances, and over the whole run the Global messages cancel (what the shards'
  septic digests add up to);
* instruction semantics against an independent model written from the RISC-V specification, on operands that include the
  division and shift edge cases (hand-assembled programs: tests/rv_asm.py), with x0 destinations (AluX0 / LoadX0 rows)."""
import hashlib
import os
import struct
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import machine_check as MC
import rv_asm as A

from sp1_amd.machines import public_values as PVM
from sp1_amd.machines import riscv_exec as X
from sp1_amd.machines import riscv_trace as RT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1


def _elf(name):
    return X.guest_file(name + ".elf")


def check_shard(machine, tabs, publics):
    """Every constraint on every row, every bus — the chips' and the public values' (eval_public_values): returns
    (failing chips, unbalanced message keys). The shard must be one of the machine's shape clusters."""
    assert frozenset(a.name for a, _ in machine) in RT.chip_clusters()
    return MC.check_shard(machine, tabs, publics, PVM.program())


def run_program(elf, stdin, max_cycles):
    ex = X.Executor(elf, stdin=stdin)
    kinds, gevs, pvs, cycles, last = [], [], [], 0, None
    for kind, machine, tabs, publics, gev, sh in X.program_shards(ex, max_cycles):
        bad, imb = check_shard(machine, tabs, publics)
        assert not bad and not imb, (kind, bad, imb)
        kinds.append(kind)
        gevs.append(gev)
        pvs.append([int(v) for v in publics])
        if sh is not None:
            cycles, last = cycles + sh.cycles, sh
            entry = sh.pc_start if len(kinds) == 1 else entry
    # the Global events of all shards cancel against the initialisation of the program's memory image: the part of the bus the
    # verifying key stands for (vk.initial_global_cumulative_sum), not a shard
    assert not X.global_events_balance(gevs + [X.image_events(ex)])
    assert X.global_events_balance(gevs)
    # what `SP1Prover::verify` checks across the shards of a core proof before it verifies each of them: the public values chain
    # (timestamps, pcs, exit codes, digests, address chains) from the entry point to HALT and the shards' septic digests add up to zero
    vk = X.verifying_key_words(ex, entry)
    err = PVM.verify_proof_public_values([pvs[i] for i in X.proof_order(kinds)], entry, vk[3:])
    if last.commit_syscall and last.commit_deferred_syscall:
        assert err is None
        assert PVM.verify_proof_public_values([pvs[i] for i in X.proof_order(kinds)], entry) == "global cumulative sum is not zero"   # (without the key's digest)
    else:       # a hand-assembled program that halts without COMMIT / COMMIT_DEFERRED_PROOFS: the reference refuses exactly that
        assert err == "prev_commit_syscall doesn't equal the previous shard's commit_syscall"
    return ex, kinds, cycles, last


GUEST_SHA256 = {   # of the files in the reference tree (bench/programs/README.md names them): the fixtures are those bytes
    "fibonacci.elf": "b37b2e0ab54497b1b0e24001a4e7d556479563dcfcc2802f77278141a3c109b4",
    "loop.elf": "daa497ce9a59cc06a014cb437457737dab0b9e27063a00c6619e814ba457677a",
    "keccak.elf": "cf37e0e70fb9f72f095f36e2b2a8a398d1463ee14f5bfc6da010128b8f72dfbf",
    "sha2.elf": "233da949bacfd62e8d9356c19acb4bbe0e42c60a5f327dc213df8efef56f8965",
    "poseidon2.elf": "c1335ee14dfd1d2ff92d8159468bc4b8f034cb6b8b556e2947c7e5942c0a72b3",
    "rsp.elf": "6077a757923815224135f8b6d7e312ceebe3dc4ce5fef57a9619a65a81ff2dd1",
    "rsp_input_21740136.bin": "b52377988c8cf68246234790c7c4cde97d670fa94d86d5f65e8501998d458e25",
}


def test_guest_fixtures_are_the_reference_binaries():
    for name, digest in GUEST_SHA256.items():
        assert hashlib.sha256(X.guest_file(name)).hexdigest() == digest, name


def _digest_words(data):
    return list(struct.unpack("<8I", hashlib.sha256(data).digest()))


def test_fibonacci_elf_runs_and_every_shard_checks():
    # `stdin.write(&n)` with n: usize (sp1-gpu/crates/perf/src/lib.rs:L25-L29): bincode = 8 little-endian bytes
    ex, kinds, cycles, last = run_program(_elf("fibonacci"), [struct.pack("<Q", 300)], 3000)
    assert kinds == ["core"] * 3 + ["memory"] and cycles > 8000
    assert last.halted and last.exit_code == 0 and last.next_pc == 1
    assert ex.output(1).startswith(b"result: ")
    assert last.commit_syscall == 1 and last.committed_value_digest == _digest_words(ex.output(0))
    # the cycle count is a function of n: 9 instructions per iteration
    ex2, _, cycles2, _ = run_program(_elf("fibonacci"), [struct.pack("<Q", 400)], 1 << 20)
    assert cycles2 - cycles == 900 and ex2.output(1) != ex.output(1)


def test_keccak_elf_with_its_precompile_shard():
    ex, kinds, cycles, last = run_program(_elf("keccak"), [bytes(300)], 4000)
    assert kinds[-2:] == ["keccak", "memory"] and set(kinds[:-2]) == {"core"}
    assert last.halted and last.exit_code == 0
    assert len(ex.output(0)) == 32 and last.committed_value_digest == _digest_words(ex.output(0))


def test_sha2_elf_with_its_two_precompile_shards():
    data = bytes(range(200))
    ex, kinds, cycles, last = run_program(_elf("sha2"), [data], 8000)
    assert kinds[-3:] == ["sha_extend", "sha_compress", "memory"] and set(kinds[:-3]) == {"core"}
    assert last.halted and last.exit_code == 0
    assert hashlib.sha256(data).digest() in ex.output(0)                 # the guest commits SHA-256 of its input, computed by the precompiles
    assert last.committed_value_digest == _digest_words(ex.output(0))    # ... and so is the digest of its public values


def test_poseidon2_elf_with_its_precompile_shards():
    ex, kinds, cycles, last = run_program(_elf("poseidon2"), [struct.pack("<Q", 20)], 1 << 20)
    assert kinds == ["core", "poseidon2", "sha_extend", "sha_compress", "memory"]
    assert last.halted and last.exit_code == 0 and b"successfully evaluated poseidon2" in ex.output(1)
    assert last.committed_value_digest == _digest_words(ex.output(0))


def test_poseidon2_system_call_against_the_host_permutation():
    """Two POSEIDON2 calls from a hand-assembled program (the second on a state the CPU touched in between): the words written are
    the library's host permutation of the words read, and every shard of the run checks."""
    import ctypes
    from sp1_amd import _lib

    def permute(words):                                                   # canonical words -> canonical words
        m = np.ascontiguousarray((words.astype(np.uint64) << np.uint64(32)) % np.uint64(MC.P), dtype=np.uint32)
        _lib.check(_lib.load().sp1hip_poseidon2_permute_host(m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 1, 1))
        return (m.astype(np.uint64) * np.uint64(MC.R_INV)) % np.uint64(MC.P)
    state = [(17 * i + 3, (1 << 30) + i) for i in range(8)]
    data = b"".join(struct.pack("<II", lo, hi) for lo, hi in state) + bytes(64)
    w = A.li(10, 0x78100000) + A.li(11, 0) + A.li(5, 0x133) + [A.enc("ecall")] + [A.enc("ld", 13, 10, 0), A.enc("sd", 13, 10, 64)] \
        + A.li(5, 0x133) + [A.enc("ecall")] + A.halt(0)
    ex, kinds, _, last = run_program(A.elf(w, data=data), [], 1 << 20)
    assert kinds == ["core", "poseidon2", "memory"] and last.exit_code == 0
    gm = {int(r[0]): int(r[2]) & M64 for r in ex.global_memory()}
    flat = np.array([x for pair in state for x in pair], dtype=np.uint32)
    once = permute(flat)
    twice = permute(once)
    assert [gm[0x78100000 + 8 * i] for i in range(8)] == [int(twice[2 * i]) | (int(twice[2 * i + 1]) << 32) for i in range(8)]
    assert gm[0x78100000 + 64] == int(once[0]) | (int(once[1]) << 32)     # the CPU copied word 0 of the first result


def test_uint256_mulmod_system_call_and_its_shard():
    """UINT256_MUL from a hand-assembled program: x <- x * y mod m for four operand sets (a zero modulus means 2^256), against
    Python's integers; the precompile shard (FieldOpCols: the byte-limb polynomial identity with its witness, FieldLtCols) checks."""
    words = lambda v: b"".join(struct.pack("<Q", (v >> (64 * i)) & M64) for i in range(4))
    cases = [(0x1234567890ABCDEF << 130 | 77, (1 << 255) + 12345, (1 << 256) - 189), (3, 5, 0), ((1 << 256) - 1, (1 << 256) - 1, 0), (7, 9, 1 << 200)]
    data, prog = b"", A.li(28, 0x78100000)
    for i, (x, y, m) in enumerate(cases):
        data += words(x) + words(y) + words(m)
        prog += [A.enc("addi", 10, 28, 96 * i), A.enc("addi", 11, 28, 96 * i + 32)] + A.li(5, 0x0001011D) + [A.enc("ecall")]
    ex, kinds, _, last = run_program(A.elf(prog + A.halt(0), data=data + bytes(32)), [], 1 << 20)
    assert kinds == ["core", "uint256", "memory"] and last.exit_code == 0
    gm = {int(r[0]): int(r[2]) & M64 for r in ex.global_memory()}
    for i, (x, y, m) in enumerate(cases):
        assert sum(gm[0x78100000 + 96 * i + 8 * k] << (64 * k) for k in range(4)) == (x * y) % (m if m else 1 << 256)


def test_secp256k1_add_and_double_system_calls_and_their_shards():
    """3G by the precompiles from a hand-assembled program: p <- 2 p (SECP256K1_DOUBLE, in place), then q <- q + p with q = G
    (SECP256K1_ADD writes its first argument), then the sum doubled again — against Python's affine arithmetic; both precompile
    shards (ten / eleven FieldOpCols per row, padding rows on the reference's dummy operands) check row by row."""
    Pm = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
    G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)

    def add(p, q):
        lam = (3 * p[0] * p[0] * pow(2 * p[1], Pm - 2, Pm) if p == q else (q[1] - p[1]) * pow(q[0] - p[0], Pm - 2, Pm)) % Pm
        x = (lam * lam - p[0] - q[0]) % Pm
        return x, (lam * (p[0] - x) - p[1]) % Pm
    words = lambda v: b"".join(struct.pack("<Q", (v >> (64 * i)) & M64) for i in range(4))
    data = words(G[0]) + words(G[1]) + words(G[0]) + words(G[1])
    prog = A.li(28, 0x78100000)
    prog += [A.enc("addi", 10, 28, 0), A.enc("addi", 11, 0, 0)] + A.li(5, 0x0000010B) + [A.enc("ecall")]          # p = 2G
    prog += [A.enc("addi", 10, 28, 64), A.enc("addi", 11, 28, 0)] + A.li(5, 0x0001010A) + [A.enc("ecall")]        # q = G + 2G
    prog += [A.enc("addi", 10, 28, 64), A.enc("addi", 11, 0, 0)] + A.li(5, 0x0000010B) + [A.enc("ecall")]         # q = 6G
    ex, kinds, _, last = run_program(A.elf(prog + A.halt(0), data=data + bytes(32)), [], 1 << 20)
    assert kinds == ["core", "secp256k1_add", "secp256k1_double", "memory"] and last.exit_code == 0
    gm = {int(r[0]): int(r[2]) & M64 for r in ex.global_memory()}
    point = lambda off: tuple(sum(gm[0x78100000 + off + 32 * c + 8 * k] << (64 * k) for k in range(4)) for c in range(2))
    g2 = add(G, G)
    g3 = add(G, g2)
    assert point(0) == g2 and point(64) == add(g3, g3)


def test_secp256k1_add_of_equal_x_is_an_executor_error():
    from sp1_amd import _lib
    words = lambda v: b"".join(struct.pack("<Q", (v >> (64 * i)) & M64) for i in range(4))
    data = (words(5) + words(7)) * 2
    prog = A.li(28, 0x78100000) + [A.enc("addi", 10, 28, 0), A.enc("addi", 11, 28, 64)] + A.li(5, 0x0001010A) + [A.enc("ecall")]
    ex = X.Executor(A.elf(prog + A.halt(0), data=data + bytes(32)), stdin=[])
    with pytest.raises(_lib.Sp1HipError, match="equal x"):
        ex.run_shard(1 << 20)


def test_a_flipped_sha_cell_is_caught():
    from sp1_amd.machines import riscv as R
    ex = X.Executor(_elf("sha2"), stdin=[bytes(10)])
    shards = list(X.program_shards(ex, 1 << 20))
    kind, machine, tabs, publics, _, _ = next(s for s in shards if s[0] == "sha_compress")
    assert check_shard(machine, tabs, publics) == ([], 0)
    col = R.chip("ShaCompress")[0].layout["temp1.value"]
    tabs["ShaCompress"][1][20, col] = (tabs["ShaCompress"][1][20, col] + 1) % MC.P
    bad, imb = check_shard(machine, tabs, publics)
    assert bad == ["ShaCompress"]


def test_rsp_elf_runs_a_whole_block():
    """The Reth block-execution client on the reference's recorded input (block 21740136), start to HALT: 9.8e7 cycles, 23k
    KECCAK_PERMUTE calls, 81k secp256k1 point operations (signature recovery; their square-root hints come from the FP_SQRT hook
    and are checked by the guest), SHA-256 of the committed block hash. A core shard from the start and one from 3e7 cycles in,
    and the first events of both curve precompiles, check row by row; the committed digest is SHA-256 of the public values."""
    from sp1_amd.machines import riscv_more_trace as MT
    data = X.guest_file("rsp_input_21740136.bin")
    ex = X.Executor(_elf("rsp"), stdin=[data])
    sh = ex.run_shard(1 << 16)
    machine, tabs, publics = X.shard_tables(ex, sh)
    assert check_shard(machine, tabs, publics) == ([], 0)
    ex.run_shard(30_000_000, record=False)
    sh = ex.run_shard(1 << 16)
    assert sh.cycles == 1 << 16 and sh.clk_start > 8 * 30_000_000
    machine, tabs, publics = X.shard_tables(ex, sh)
    assert check_shard(machine, tabs, publics) == ([], 0)
    assert {"LoadByte", "StoreDouble", "Bitwise", "Global"} <= {a.name for a, _ in machine}
    cycles, counts, first = 30_000_000 + (2 << 16), {"keccak": 0, "secp256k1_add": 0, "secp256k1_double": 0, "sha_compress": 0}, {}
    while not sh.halted:
        sh = ex.run_shard(1 << 22, copy=False)
        cycles += sh.cycles
        for k in counts:
            ev = getattr(sh, k)
            counts[k] += ev.shape[0]
            if ev.shape[0] and k not in first:
                first[k] = np.array(ev[:40])
    assert sh.exit_code == 0 and cycles == 98_339_210
    # (668 more KECCAK_PERMUTE calls in the 3e7 cycles that ran without recording: 23254 in all)
    assert counts == {"keccak": 22586, "secp256k1_add": 26874, "secp256k1_double": 53760, "sha_compress": 12}
    assert sh.committed_value_digest == _digest_words(ex.output(0))
    for kind, build in (("secp256k1_add", MT.secp256k1_add_shard_from), ("secp256k1_double", MT.secp256k1_double_shard_from)):
        machine, tabs, publics, _ = build(first[kind])
        assert check_shard(machine, tabs, publics)[0] == []          # a slice of the run: its Global messages balance only with the rest
    assert X.split_thresholds(ex.program()[1].shape[0])["keccak"] < counts["keccak"]        # the Keccak calls do not fit one shard


def test_precompile_and_memory_events_split_into_shards_at_the_thresholds(monkeypatch):
    """With the thresholds of `SplitOpts::new` shrunk (2 Keccak calls, 96 addresses per shard) the keccak guest's run has several
    precompile and memory shards; each checks on its own — the memory shards chained through previous_init_addr — and the Global
    messages of all of them still cancel. The real thresholds for this program are those of the reference's cost model."""
    real = X.split_thresholds(4096)
    assert real["keccak"] % 32 == 0 and 4000 < real["keccak"] < 6000 and real["secp256k1_double"] > real["secp256k1_add"] > 30000
    assert real["memory"] % 32 == 0 and real["memory"] <= (1 << 22) // 2
    monkeypatch.setattr(X, "split_thresholds", lambda rows: dict(real, keccak=2, memory=96))
    ex, kinds, _, last = run_program(_elf("keccak"), [bytes(300)], 1 << 20)
    assert kinds.count("keccak") >= 2 and kinds.count("memory") >= 2 and last.exit_code == 0


def test_global_message_balance_of_a_large_run_uses_fingerprints_and_still_names_what_is_missing():
    rng = np.random.default_rng(5)
    n = 300_000
    msg = rng.integers(0, 1 << 30, size=(n, 8))
    kind = rng.integers(1, 12, size=(n, 1))
    one, zero = np.ones((n, 1), dtype=np.int64), np.zeros((n, 1), dtype=np.int64)
    sends = torch.as_tensor(np.concatenate([msg, one, zero, kind], axis=1))
    recvs = torch.as_tensor(np.concatenate([msg, zero, one, kind], axis=1)[rng.permutation(n)])
    assert X.global_events_balance([sends, recvs]) == []
    recvs[12345, 3] += 1
    assert len(X.global_events_balance([sends, recvs])) == 2           # the message that was never received, the one never sent


def test_core_shards_are_cut_by_the_trace_area_estimator():
    """`cut_by_area`: the executor ends a shard where the reference's ShapeChecker would — when the estimated trace area reaches the
    element threshold less the HALT allowance. With the threshold set 2e6 cells above the fixed tables the fibonacci guest's run
    is cut into several shards whose REAL tables (what the tracer builds, padded to 32 rows) stay below the threshold and within a
    few per cent of the estimate; every shard still checks and the Global messages cancel. No cut between COMMIT and HALT."""
    from sp1_amd.machines import riscv as R
    cost = lambda name: (lambda a: a.main_width + a.prep_width)(R.chip(name)[0])
    ex = X.Executor(_elf("fibonacci"), stdin=[struct.pack("<Q", 10000)])
    rows = ex.program()[1].shape[0]
    fixed = -(-rows // 32) * 32 * cost("Program") + (1 << 16) * cost("Byte") + (1 << 17) * cost("Range")
    threshold = fixed + 2_000_000
    ex.cut_by_area(element_threshold=threshold)
    kinds, gevs, sizes = [], [], []
    for kind, machine, tabs, publics, gev, sh in X.program_shards(ex, 1 << 40):
        assert check_shard(machine, tabs, publics) == ([], 0)
        kinds.append(kind)
        gevs.append(gev)
        if kind == "core":
            real = sum(int(tabs[a.name][1].shape[0]) * (a.main_width + a.prep_width) for a, _ in machine)
            sizes.append((sh.cycles, sh.estimated_area, real, sh.halted))
    assert not X.global_events_balance(gevs + [X.image_events(ex)])
    assert kinds.count("core") >= 3
    for cycles, est, real, halted in sizes:
        assert real <= threshold and abs(est - real) < 0.25 * (threshold - fixed) + (1 << 19)
        if not halted:
            assert threshold - (1 << 18) <= est < threshold - (1 << 18) + (1 << 18)        # stopped at the first instruction past the limit
    assert sizes[-1][3]
    # cycle counts alone again
    from sp1_amd import _lib
    _lib.check(ex.lib.sp1hip_rv64_set_shard_limits(ex.h, None))


def test_an_unknown_hook_is_an_executor_error():
    from sp1_amd import _lib
    prog = A.li(10, 15) + A.li(11, 0x78100000) + A.li(12, 8) + A.li(5, 2) + [A.enc("ecall")]       # WRITE(fd 15 = ecrecover hook, buf, 8)
    ex = X.Executor(A.elf(prog + A.halt(0), data=bytes(32)), stdin=[])
    with pytest.raises(_lib.Sp1HipError, match="hook"):
        ex.run_shard(1 << 20)


def test_loop_elf():
    ex, kinds, cycles, last = run_program(_elf("loop"), [struct.pack("<Q", 500)], 1 << 20)
    assert kinds == ["core", "memory"] and last.exit_code == 0
    assert last.committed_value_digest == _digest_words(ex.output(0))


def test_guest_panic_is_an_exit_code_not_an_executor_error():
    ex = X.Executor(_elf("loop"), stdin=[b"\x01\x02\x03"])                # too short for the usize the guest deserialises
    shards = list(ex.shards(1 << 20))
    assert shards[-1].halted and shards[-1].exit_code == 1 and b"panicked" in ex.output(1)


def test_fast_bus_check_sees_a_missing_message():
    ex = X.Executor(_elf("fibonacci"), stdin=[struct.pack("<Q", 5)])
    sh = ex.run_shard(1 << 20)
    machine, tabs, publics = X.shard_tables(ex, sh)
    assert check_shard(machine, tabs, publics) == ([], 0)
    main = tabs["Add"][1]
    main[0, R_LAYOUT("Add")["value"]] = (main[0, R_LAYOUT("Add")["value"]] + 1) % MC.P       # a wrong sum: constraint AND memory bus
    bad, imb = check_shard(machine, tabs, publics)
    assert bad == ["Add"] and imb > 0


def R_LAYOUT(chip):
    from sp1_amd.machines import riscv as R
    return R.chip(chip)[0].layout


# ---------------------------------------------------------------------------------------------------------------------
# instruction semantics against the specification
def _s(v, bits=64):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def _tdiv(a, b):
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def spec(op, b, c):
    """rd of `op rd, rs1 = b, rs2 = c` per the unprivileged specification (chapters RV64I and M), as a u64."""
    sb, sc, wb, wc = _s(b), _s(c), _s(b, 32), _s(c, 32)
    sx = lambda v: _s(v, 32) & M64
    f = {
        "add": lambda: b + c, "sub": lambda: b - c, "xor": lambda: b ^ c, "or": lambda: b | c, "and": lambda: b & c,
        "sll": lambda: b << (c & 63), "srl": lambda: b >> (c & 63), "sra": lambda: sb >> (c & 63),
        "slt": lambda: int(sb < sc), "sltu": lambda: int(b < c),
        "mul": lambda: b * c, "mulh": lambda: (sb * sc) >> 64, "mulhu": lambda: (b * c) >> 64, "mulhsu": lambda: (sb * c) >> 64,
        "div": lambda: -1 if c == 0 else _tdiv(sb, sc), "divu": lambda: M64 if c == 0 else b // c,
        "rem": lambda: sb if c == 0 else sb - sc * _tdiv(sb, sc), "remu": lambda: b if c == 0 else b % c,
        "addw": lambda: sx(b + c), "subw": lambda: sx(b - c), "mulw": lambda: sx(b * c),
        "sllw": lambda: sx(b << (c & 31)), "srlw": lambda: sx((b & 0xFFFFFFFF) >> (c & 31)), "sraw": lambda: sx(wb >> (c & 31)),
        "divw": lambda: -1 if wc == 0 else sx(_tdiv(wb, wc)), "divuw": lambda: M64 if wc == 0 else sx((b & 0xFFFFFFFF) // (c & 0xFFFFFFFF)),
        "remw": lambda: sx(wb) if wc == 0 else sx(wb - wc * _tdiv(wb, wc)),
        "remuw": lambda: sx(wb) if wc == 0 else sx((b & 0xFFFFFFFF) % (c & 0xFFFFFFFF)),
    }[op]
    return f() & M64


EDGE = [0, 1, 2, M64, 1 << 63, (1 << 63) - 1, 1 << 31, (1 << 31) - 1, 0xFFFFFFFF, 0x80000000_00000001, 7, (-7) & M64, 63, 64, 31, 32]


def test_alu_semantics_against_the_specification_and_their_tables():
    rng = np.random.default_rng(5)
    ops = sorted(A.R_OPS)
    cases = [(op, b, c) for op in ops for b, c in [(EDGE[i], EDGE[j]) for i, j in rng.integers(0, len(EDGE), (6, 2))]]
    cases += [(op, int(rng.integers(0, 1 << 63)) * 2 + 1, int(rng.integers(0, 1 << 63)) * 2) for op in ops for _ in range(2)]
    cases += [("div", 1 << 63, M64), ("rem", 1 << 63, M64), ("divw", 1 << 31, 0xFFFFFFFF), ("remw", 1 << 31, 0xFFFFFFFF), ("divu", 5, 0),
              ("div", 5, 0), ("remuw", 5, 0)]
    words, want = [], []
    for i, (op, b, c) in enumerate(cases):
        rd = 0 if i % 11 == 10 else 13 + i % 5                            # some results are discarded into x0: AluX0 rows
        words += A.li(6, b) + A.li(7, c) + [A.enc(op, rd, 6, 7)]
        if rd:                                                           # keep the result: store it
            words += [A.enc("sd", rd, 28, 8 * len(want))]
            want.append(spec(op, b, c))
    prologue = A.li(28, 0x78100000)
    words = prologue + words + A.halt(0)
    ex = X.Executor(A.elf(words, data=bytes(8 * len(want) + 8)))
    sh = ex.run_shard(1 << 20)
    assert sh.halted and sh.exit_code == 0
    gm = {int(r[0]): int(r[2]) & M64 for r in ex.global_memory()}
    got = [gm[0x78100000 + 8 * i] for i in range(len(want))]
    wrong = [(cases[i], hex(g), hex(w)) for i, (g, w) in enumerate(zip(got, want)) if g != w]
    assert not wrong, wrong[:5]
    machine, tabs, publics = X.shard_tables(ex, sh)
    names = {a.name for a, _ in machine}
    assert {"AluX0", "DivRem", "Mul", "ShiftLeft", "ShiftRight", "Lt", "Addw", "Subw", "StoreDouble"} <= names
    assert check_shard(machine, tabs, publics) == ([], 0)


def test_loads_stores_branches_jumps_and_their_tables():
    w = A.li(28, 0x78100000) + A.li(6, 0x8877665544332211) + [
        A.enc("sd", 6, 28, 0), A.enc("sb", 6, 28, 9), A.enc("sh", 6, 28, 18), A.enc("sw", 6, 28, 28),
        A.enc("lb", 13, 28, 7), A.enc("lbu", 14, 28, 7), A.enc("lh", 15, 28, 6), A.enc("lhu", 16, 28, 6), A.enc("lw", 17, 28, 4),
        A.enc("lwu", 18, 28, 4), A.enc("ld", 19, 28, 0), A.enc("ld", 0, 28, 8), A.enc("lbu", 0, 28, 1),          # LoadX0 rows
        A.enc("sd", 13, 28, 32), A.enc("sd", 14, 28, 40), A.enc("sd", 15, 28, 48), A.enc("sd", 16, 28, 56), A.enc("sd", 17, 28, 64),
        A.enc("sd", 18, 28, 72), A.enc("sd", 19, 28, 80),
        A.enc("blt", 13, 0, 8), A.enc("addi", 20, 0, 1),                  # taken: skips the addi
        A.enc("bgeu", 13, 0, 8), A.enc("addi", 21, 0, 2),                 # taken (unsigned): skips
        A.enc("beq", 13, 14, 8), A.enc("addi", 22, 0, 3),                 # not taken
        A.enc("jal", 1, 8), A.enc("addi", 23, 0, 4),                      # skips; x1 = pc + 4
        A.enc("auipc", 24, 0), A.enc("jalr", 0, 24, 12), A.enc("addi", 25, 0, 5),      # jumps over the addi, rd = x0
        A.enc("sd", 20, 28, 88), A.enc("sd", 22, 28, 96), A.enc("sd", 23, 28, 104), A.enc("sd", 25, 28, 112),
    ] + A.halt(0)
    ex = X.Executor(A.elf(w, data=bytes(128)))
    sh = ex.run_shard(1 << 20)
    assert sh.halted and sh.exit_code == 0
    gm = {int(r[0]): int(r[2]) & M64 for r in ex.global_memory()}
    mem = lambda off: gm[0x78100000 + off]
    assert mem(0) == 0x8877665544332211 and mem(8) == 0x1100 and mem(16) == 0x2211_0000 and mem(24) == 0x44332211_00000000
    assert [mem(32 + 8 * i) for i in range(7)] == [0xFFFFFFFFFFFFFF88, 0x88, 0xFFFFFFFFFFFF8877, 0x8877, 0xFFFFFFFF88776655, 0x88776655,
                                                   0x8877665544332211]
    assert [mem(88), mem(96), mem(104), mem(112)] == [0, 3, 0, 0]
    machine, tabs, publics = X.shard_tables(ex, sh)
    assert {"LoadX0", "LoadByte", "LoadHalf", "LoadWord", "StoreByte", "StoreHalf", "StoreWord", "Branch", "Jal", "Jalr", "UType"} <= {a.name for a, _ in machine}
    assert check_shard(machine, tabs, publics) == ([], 0)


def test_executor_errors_are_reported_not_swallowed():
    from sp1_amd import _lib
    with pytest.raises(_lib.Sp1HipError):
        X.Executor(b"not an elf" * 10)
    ex = X.Executor(A.elf(A.li(28, 0x78100001) + [A.enc("ld", 5, 28, 0)] + A.halt(0), data=bytes(16)))      # misaligned load
    with pytest.raises(_lib.Sp1HipError, match="misaligned"):
        ex.run_shard(100)
    ex = X.Executor(A.elf(A.li(5, 0x0000_010C) + [A.enc("ecall")] + A.halt(0)))                               # SECP256K1_DECOMPRESS: not implemented
    with pytest.raises(_lib.Sp1HipError, match="0x10c"):
        ex.run_shard(100)


def test_the_oracle_proves_and_verifies_the_halting_shard_with_its_public_values():
    """The shard that executes COMMIT x 8, COMMIT_DEFERRED_PROOFS x 8 and HALT reads the committed digest, the exit code and the
    commit flags from its public values: the pinned verifier accepts the proof with the shard's own values and rejects it when
    one digest byte is different."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    from sp1_amd.machines import riscv_trace as RT
    ex = X.Executor(_elf("fibonacci"), stdin=[struct.pack("<Q", 20)])
    sh = ex.run_shard(1 << 20)
    assert sh.halted
    machine, tabs, publics = X.shard_tables(ex, sh)
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None) for a, i in machine]
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None) for a, i in machine]
    L, lsh, batch = 17, 12, 8
    prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, 1)
    orc.set_gkr_sparse(True)
    try:
        for flip, want_ok in ((None, True), (X.PV["committed_value_digest"] + 5, False)):
            pv = publics.clone()
            if flip is not None:
                pv[flip] = (pv[flip] + 1) % 256
            ch = orc.Challenger()
            ch.observe(prep.commit)
            v_ch = ch.clone()
            blob = orc.shard_prove(host, RT.to_monty_np(pv), prep, L, lsh, batch, ch, 1, 5, 4)
            assert (orc.shard_verify(shapes, prep.commit, blob, L, lsh, v_ch, 1, 5, 4, pv_program=PVM.verifier_program()) == 0) == want_ok
    finally:
        orc.set_gkr_sparse(False)


def test_memory_image_follows_the_reference_elf_rules():
    """`Program::memory_image` (disassembler/elf.rs:L320-L353): every 8-byte word of every PT_LOAD segment, file bytes and zero
    fill alike, by address — and the reference's zero-fill rule as it is written: `image.insert(addr - addr % 8, 0)` REPLACES the
    word, so a segment whose file part ends in the middle of a word loses that word's file half. The executor's memory starts as
    the image says (the verifying key's digest is computed from it: a loader with other rules would prove another statement)."""
    import struct as st
    import rv_asm as A
    data = bytes(range(1, 13))                                            # 12 file bytes: one whole word and a half
    elf = bytearray(A.elf(A.li(5, 0) + A.halt(0), data=data))
    ph = 64 + 56                                                          # the second program header: p_filesz at +32, p_memsz at +40
    assert st.unpack_from("<Q", elf, ph + 32)[0] == 12
    st.pack_into("<Q", elf, ph + 40, 28)                                  # memsz: zero fill up to byte 28
    ex = X.Executor(bytes(elf), stdin=[])
    img = {int(a): int(v) & ((1 << 64) - 1) for a, v in ex.memory_image()}
    base = 0x78100000
    assert img[base] == int.from_bytes(data[:8], "little")
    assert img[base + 8] == 0 and img[base + 16] == 0 and img[base + 24] == 0   # the half word 9..12 is gone with the first zero fill
    assert base + 32 not in img
    text = [a for a in img if a < base]
    assert text == sorted(text) and len(text) == (len(A.li(5, 0) + A.halt(0)) + 1) // 2
    ev = X.image_events(ex)
    assert ev.shape == (len(img), 11) and int(ev[:, 8].sum()) == len(img) and int(ev[:, 9].sum()) == 0      # one SEND per word


def test_hinted_words_are_initialised_whether_or_not_they_are_read():
    """HINT_READ writes whole words and a tail word only if there is one (minimal/postprocess.rs:L10-L36); every hinted word is
    initialised with its hint and finalised (controller/global.rs:L133-L142), read or not."""
    import rv_asm as A
    buf = 0x78200000
    prog = A.li(10, buf) + A.li(11, 16) + A.li(5, 0xF1) + [A.enc("ecall")]           # HINT_READ(buf, 16): two words, no tail
    prog += A.li(10, buf + 64) + A.li(11, 5) + A.li(5, 0xF1) + [A.enc("ecall")]      # HINT_READ(buf + 64, 5): a tail word only
    prog += [A.enc("ld", 6, 10, 0)]                                                 # the program reads only the tail word
    ex = X.Executor(A.elf(prog + A.halt(0)), stdin=[bytes(range(16)), b"\x01\x02\x03\x04\x05"])
    while not ex.halted:
        ex.run_shard(1 << 20)
    gm = {int(r[0]): (int(r[1]) & ((1 << 64) - 1), int(r[2]) & ((1 << 64) - 1), int(r[3])) for r in ex.global_memory()}
    w0, w1 = int.from_bytes(bytes(range(8)), "little"), int.from_bytes(bytes(range(8, 16)), "little")
    assert gm[buf] == (w0, w0, 0) and gm[buf + 8] == (w1, w1, 0)            # never read: initial = final = the hint, timestamp 0
    assert buf + 16 not in gm                                               # no empty tail word
    tail = int.from_bytes(b"\x01\x02\x03\x04\x05", "little")
    assert gm[buf + 64][:2] == (tail, tail) and gm[buf + 64][2] > 0          # read: it carries the load's timestamp
