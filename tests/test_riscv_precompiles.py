"""The field / curve precompiles beyond secp256k1 ("every RiscvAir chip"): secp256r1 / bn254 / bls12-381 point
addition and doubling, the Fp / Fp2 tower operations of bn254 and bls12-381, ed25519 addition and decompression, 256-bit add / mul
with carry. For each: the chip's column / constraint counts equal the reference's cost tables (tests/test_riscv_more.py does that for
every entry of MORE_RECORDED); here a hand-assembled program calls them, the executor's results equal Python's integers, and every
shard of the run — the core shard with its ECALLs, one precompile shard per chip, the memory shard — checks row by row with the
Global messages cancelling. Operands and expected values come from Python arithmetic written in this file, not from the executor."""
import struct

import pytest

import rv_asm as A
from sp1_amd import _lib
from sp1_amd.machines import riscv_exec as X
from sp1_amd.machines import riscv_more as M
from test_riscv_exec import M64, run_program

DATA = 0x78100000
SECP256R1_G = (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296, 0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)
BN254_G = (1, 2)
BLS12381_G = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
              0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
ED_B = (15112221349535400772501151409588531511454012693041857206046113283949847762202, 46316835694926478169428394003475163141307993866256225615783033603165251855960)


def words(v, n):
    return b"".join(struct.pack("<Q", (v >> (64 * i)) & M64) for i in range(n))


def call(code, a0, a1):
    return [A.enc("addi", 10, 28, a0), A.enc("addi", 11, 28, a1) if a1 is not None else A.enc("addi", 11, 0, 0)] + A.li(5, code) + [A.enc("ecall")]


def memory_of(ex):
    return {int(r[0]): int(r[2]) & M64 for r in ex.global_memory()}


def value_at(gm, off, n):
    return sum(gm[DATA + off + 8 * k] << (64 * k) for k in range(n))


def w_add(p, q, Pm, a):
    lam = ((3 * p[0] * p[0] + a) * pow(2 * p[1], Pm - 2, Pm) if p == q else (q[1] - p[1]) * pow(q[0] - p[0], Pm - 2, Pm)) % Pm
    x = (lam * lam - p[0] - q[0]) % Pm
    return x, (lam * (p[0] - x) - p[1]) % Pm


@pytest.mark.parametrize("curve, G, add_code, double_code", [("Secp256r1", SECP256R1_G, 0x0001012C, 0x0000012D), ("Bn254", BN254_G, 0x0001010E, 0x0000010F),
                                                             ("Bls12381", BLS12381_G, 0x0001011E, 0x0000011F)])
def test_weierstrass_add_and_double(curve, G, add_code, double_code):
    """p <- 2 p (in place), then q <- q + p with q = G, then q <- 2 q: 2G and 6G against affine arithmetic over Python integers."""
    Pm, a, nl = M.CURVES[curve][:3]
    n = nl // 8
    b_coeff = {"Secp256r1": 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B, "Bn254": 3, "Bls12381": 4}[curve]
    assert (G[1] * G[1] - G[0] ** 3 - a * G[0] - b_coeff) % Pm == 0               # the generator is on its curve
    pt = words(G[0], n) + words(G[1], n)
    size = 2 * 8 * n
    prog = A.li(28, DATA) + call(double_code, 0, None) + call(add_code, size, 0) + call(double_code, size, None)
    ex, kinds, _, last = run_program(A.elf(prog + A.halt(0), data=pt + pt + bytes(32)), [], 1 << 20)
    assert kinds == ["core", curve.lower() + "_add", curve.lower() + "_double", "memory"] and last.exit_code == 0
    gm = memory_of(ex)
    g2 = w_add(G, G, Pm, a)
    g3 = w_add(G, g2, Pm, a)
    assert (value_at(gm, 0, n), value_at(gm, 8 * n, n)) == g2
    assert (value_at(gm, size, n), value_at(gm, size + 8 * n, n)) == w_add(g3, g3, Pm, a)


@pytest.mark.parametrize("field", ["Bn254", "Bls12381"])
def test_fp_and_fp2_operations(field):
    """x <- x op y over Fp (add, sub, mul; once with x and y the same words: a squaring) and over Fp2 = Fp[u] / (u^2 + 1) (add, sub,
    mul). The three Fp calls share one chip and one shard, the Fp2 add / sub calls another, the Fp2 products a third."""
    Pm, nl = M.FP_FIELDS[field][:2]
    n = nl // 8
    codes = [0x00010100 | c for c in M.FP_SYSCALLS[field]]
    a, b = (3 ** 200 + 12345) % Pm, Pm - 7
    c0, c1, d0, d1 = 5 ** 150 % Pm, Pm - 1, 7 ** 130 % Pm, 11 ** 100 % Pm
    slot = 8 * n
    data = words(a, n) + words(b, n) + words(a, n) + words(a, n) + words(b, n) + words(c0, n) + words(c1, n) + words(d0, n) + words(d1, n)
    data += words(c0, n) + words(c1, n) + words(c0, n) + words(c1, n)
    prog = A.li(28, DATA)
    prog += call(codes[0], 0, slot)                # [0] = a + b
    prog += call(codes[2], 2 * slot, 2 * slot)     # [2] = a * a
    prog += call(codes[1], 3 * slot, 4 * slot)     # [3] = a - b
    prog += call(codes[2], 4 * slot, 0)            # [4] = b * (a + b)
    prog += call(codes[5], 5 * slot, 7 * slot)     # [5, 6] = (c0 + c1 u)(d0 + d1 u)
    prog += call(codes[3], 9 * slot, 7 * slot)     # [9, 10] = c + d
    prog += call(codes[4], 11 * slot, 7 * slot)    # [11, 12] = c - d
    ex, kinds, _, last = run_program(A.elf(prog + A.halt(0), data=data + bytes(32)), [], 1 << 20)
    f = field.lower()
    assert kinds == ["core", f + "_fp", f + "_fp2_addsub", f + "_fp2_mul", "memory"] and last.exit_code == 0
    gm = memory_of(ex)
    at = lambda k: value_at(gm, k * slot, n)
    assert at(0) == (a + b) % Pm and at(2) == a * a % Pm and at(3) == (a - b) % Pm and at(4) == b * ((a + b) % Pm) % Pm
    assert (at(5), at(6)) == ((c0 * d0 - c1 * d1) % Pm, (c0 * d1 + c1 * d0) % Pm)
    assert (at(9), at(10)) == ((c0 + d0) % Pm, (c1 + d1) % Pm) and (at(11), at(12)) == ((c0 - d0) % Pm, (c1 - d1) % Pm)


def ed_add(p, q):
    Pm, D = M.ED25519_P, M.ED25519_D
    f = D * p[0] * q[0] * p[1] * q[1] % Pm
    return ((p[0] * q[1] + q[0] * p[1]) * pow(1 + f, Pm - 2, Pm) % Pm, (p[1] * q[1] + p[0] * q[0]) * pow(1 - f, Pm - 2, Pm) % Pm)


def test_ed25519_add_and_decompress():
    """2B and 3B by ED_ADD (the complete twisted-Edwards law: adding a point to itself is an ordinary call), then ED_DECOMPRESS
    recovers x(3B) from its y coordinate with the sign bit of the true x, and from y(2B) with the OTHER sign bit the x of -2B."""
    Pm, D = M.ED25519_P, M.ED25519_D
    assert (-ED_B[0] ** 2 + ED_B[1] ** 2 - 1 - D * ED_B[0] ** 2 * ED_B[1] ** 2) % Pm == 0
    b2 = ed_add(ED_B, ED_B)
    b3 = ed_add(b2, ED_B)
    pt = words(ED_B[0], 4) + words(ED_B[1], 4)
    data = pt + pt + pt + bytes(32) + words(b3[1], 4) + bytes(32) + words(b2[1], 4)
    prog = A.li(28, DATA) + call(0x00010107, 0, 64)                               # [0] = B + B
    prog += call(0x00010107, 128, 0)                                             # [128] = B + 2B
    prog += [A.enc("addi", 10, 28, 192), A.enc("addi", 11, 0, b3[0] & 1)] + A.li(5, 0x00000108) + [A.enc("ecall")]
    prog += [A.enc("addi", 10, 28, 256), A.enc("addi", 11, 0, 1 - (b2[0] & 1))] + A.li(5, 0x00000108) + [A.enc("ecall")]
    ex, kinds, _, last = run_program(A.elf(prog + A.halt(0), data=data + bytes(32)), [], 1 << 20)
    assert kinds == ["core", "ed_add", "ed_decompress", "memory"] and last.exit_code == 0
    gm = memory_of(ex)
    assert (value_at(gm, 0, 4), value_at(gm, 32, 4)) == b2 and (value_at(gm, 128, 4), value_at(gm, 160, 4)) == b3
    assert value_at(gm, 192, 4) == b3[0] and value_at(gm, 256, 4) == Pm - b2[0] and {b3[0] & 1, 1 - (b2[0] & 1)} == {0, 1}


def test_ed_decompress_of_a_non_point_is_an_executor_error():
    data = bytes(32) + words(2, 4)                                               # y = 2: (y^2 - 1) / (d y^2 + 1) is not a square
    assert pow((4 - 1) * pow(M.ED25519_D * 4 + 1, M.ED25519_P - 2, M.ED25519_P) % M.ED25519_P, (M.ED25519_P - 1) // 2, M.ED25519_P) != 1
    prog = A.li(28, DATA) + [A.enc("addi", 10, 28, 0), A.enc("addi", 11, 0, 0)] + A.li(5, 0x00000108) + [A.enc("ecall")]
    ex = X.Executor(A.elf(prog + A.halt(0), data=data + bytes(32)), stdin=[])
    with pytest.raises(_lib.Sp1HipError, match="not a point"):
        ex.run_shard(1 << 20)


def test_uint256_add_and_mul_with_carry():
    """d, e <- low, high of a + b + c and of a * b + c (pointers c, d, e in x12, x13, x14); the second call writes d over its own
    operand a. Both system calls share the Uint256Ops chip and its shard."""
    top = (1 << 256) - 1
    cases = [(0x00010130, top, top, top), (0x00010131, top, top - 5, top), (0x00010131, 3 ** 150, 5 ** 100, 7), (0x00010130, 1, 2, 3)]
    data, prog = b"", A.li(28, DATA)
    for i, (code, a, b, c) in enumerate(cases):
        base = 160 * i
        data += words(a, 4) + words(b, 4) + words(c, 4) + bytes(64)
        d_off = base if i == 1 else base + 96                                     # case 1: d overwrites a
        prog += [A.enc("addi", 12, 28, base + 64), A.enc("addi", 13, 28, d_off), A.enc("addi", 14, 28, base + 128)] + call(code, base, base + 32)
    ex, kinds, _, last = run_program(A.elf(prog + A.halt(0), data=data + bytes(32)), [], 1 << 20)
    assert kinds == ["core", "uint256_ops", "memory"] and last.exit_code == 0
    gm = memory_of(ex)
    for i, (code, a, b, c) in enumerate(cases):
        base = 160 * i
        full = (a * b if code & 0xFF == 0x31 else a + b) + c
        assert value_at(gm, base if i == 1 else base + 96, 4) == full & top and value_at(gm, base + 128, 4) == full >> 256


def test_unreduced_operands_and_degenerate_points_stop_the_run():
    Pm = M.BN254_P
    for code, data, msg in ((0x00010128, words(Pm, 4) + words(1, 4), "not reduced"), (0x0001010E, (words(1, 4) + words(2, 4)) * 2, "equal x"),
                            (0x0000010F, words(1, 4) + words(0, 4), "y = 0")):
        prog = A.li(28, DATA) + call(code, 0, None if code == 0x0000010F else 64 if code == 0x0001010E else 32)
        ex = X.Executor(A.elf(prog + A.halt(0), data=data + bytes(32)), stdin=[])
        with pytest.raises(_lib.Sp1HipError, match=msg):
            ex.run_shard(1 << 20)


def test_pointers_the_address_operation_cannot_constrain_stop_the_run():
    """SyscallAddrOperation needs an aligned address above 2^16 (the registers live below): x10 = x11 = 0 is refused, not read."""
    ex = X.Executor(A.elf(A.li(5, 0x00010107) + [A.enc("ecall")] + A.halt(0)), stdin=[])
    with pytest.raises(_lib.Sp1HipError, match="ED_ADD arguments"):
        ex.run_shard(1 << 20)


def test_a_flipped_field_limb_is_caught():
    """One byte limb of a product changed in the Fp chip's table: the polynomial identity (and the byte range bus) no longer hold."""
    from sp1_amd.machines import riscv as R
    from sp1_amd.machines import riscv_more_trace as MT
    from test_riscv_exec import check_shard
    Pm = M.BN254_P
    prog = A.li(28, DATA) + call(0x00010128, 0, 32)
    ex = X.Executor(A.elf(prog + A.halt(0), data=words(12345, 4) + words(Pm - 2, 4) + bytes(32)), stdin=[])
    shard = ex.run_shard(1 << 20)
    machine, tabs, publics, _ = MT.family_shard_from("bn254_fp", shard.families["bn254_fp"])
    bad, imb = check_shard(machine, tabs, publics)
    assert not bad and not imb
    air = R.chip("Bn254FpOpAssign")[0]
    tabs["Bn254FpOpAssign"][1][0, air.layout["output.result"] + 3] ^= 1
    bad, imb = check_shard(machine, tabs, publics)
    assert "Bn254FpOpAssign" in bad


def test_the_oracle_proves_and_verifies_the_carry_shard():
    """One of the new shard kinds through the whole (CPU) prover: LogUp-GKR over the chip's 301 interactions (its three register
    reads among them), zerocheck over its 297 constraints, the jagged / BaseFold opening — and the pinned verifier accepts. (The
    curve shards take 20 - 50 s each in the oracle: run by hand)"""
    import numpy as np
    import pyoracle as orc
    from sp1_amd.machines import riscv_trace as RT
    top = (1 << 256) - 1
    data = words(top, 4) + words(top - 5, 4) + words(top, 4) + bytes(64)
    prog = A.li(28, DATA) + [A.enc("addi", 12, 28, 64), A.enc("addi", 13, 28, 96), A.enc("addi", 14, 28, 128)] + call(0x00010131, 0, 32)
    ex = X.Executor(A.elf(prog + A.halt(0), data=data + bytes(32)), stdin=[])
    shards = {kind: (machine, tabs, publics) for kind, machine, tabs, publics, _, _ in X.program_shards(ex, 1 << 20)}
    machine, tabs, publics = shards["uint256_ops"]
    host = [(a, i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None) for a, i in machine]
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None) for a, i in machine]
    L, lsh, batch = 17, 12, 8
    prep = orc.JaggedRound([c[3] for c in host if c[3] is not None], L, lsh, batch, 1)
    orc.set_gkr_sparse(True)
    try:
        ch = orc.Challenger()
        ch.observe(prep.commit)
        v_ch = ch.clone()
        blob = orc.shard_prove(host, RT.to_monty_np(publics), prep, L, lsh, batch, ch, 1, 5, 4)
        from sp1_amd.machines import public_values as PVM
        assert orc.shard_verify(shapes, prep.commit, blob, L, lsh, v_ch, 1, 5, 4, pv_program=PVM.verifier_program()) == 0
    finally:
        orc.set_gkr_sparse(False)


@pytest.mark.parametrize("curve", ["Bls12381", "Ed25519"])
def test_a_scalar_multiplication_by_the_precompiles(curve):
    """k G by double-and-add, every step a precompile call (bls12-381: DOUBLE in place and ADD of the base point; ed25519: ED_ADD
    for both, the law being complete), k a 40-bit scalar: ~60 calls with operands nobody chose by hand. The result equals Python's
    own scalar multiplication and both precompile shards (up to 2,399 columns, 64 rows) check row by row."""
    k = 0xB5C3A7910F
    bits = bin(k)[3:]                                                           # below the leading one
    if curve == "Ed25519":
        n, G, add, dbl = 4, ED_B, ed_add, lambda p: ed_add(p, p)
    else:
        Pm, a, nl = M.CURVES[curve][:3]
        n, G = nl // 8, BLS12381_G
        add, dbl = (lambda p, q: w_add(p, q, Pm, a)), (lambda p: w_add(p, p, Pm, a))
    size = 16 * n
    pt = words(G[0], n) + words(G[1], n)
    prog, want = A.li(28, DATA), G                                              # accumulator at +0, base point at +size, scratch copy at +2 size
    for bit in bits:
        if curve == "Ed25519":                                                  # acc <- acc + acc needs a second copy of acc: q is read, p rewritten
            for i in range(2 * n):
                prog += [A.enc("ld", 6, 28, 8 * i), A.enc("sd", 6, 28, 2 * size + 8 * i)]
            prog += call(0x00010107, 0, 2 * size)
        else:
            prog += call(0x0000011F, 0, None)
        want = dbl(want)
        if bit == "1":
            prog += call(0x00010107 if curve == "Ed25519" else 0x0001011E, 0, size)
            want = add(want, G)
    ex, kinds, _, last = run_program(A.elf(prog + A.halt(0), data=pt + pt + bytes(size) + bytes(32)), [], 1 << 20)
    assert last.exit_code == 0 and kinds == (["core", "ed_add", "memory"] if curve == "Ed25519" else ["core", "bls12381_add", "bls12381_double", "memory"])
    gm = memory_of(ex)
    assert (value_at(gm, 0, n), value_at(gm, 8 * n, n)) == want
    # the same point by an independent route: Python's pow-free ladder from the other end
    acc, addend, kk = None, G, k
    while kk:
        if kk & 1:
            acc = addend if acc is None else add(acc, addend)
        addend, kk = dbl(addend), kk >> 1
    assert acc == want


def test_an_unknown_family_is_an_error_not_a_read():
    import ctypes as C
    ex = X.Executor(A.elf(A.halt(0)), stdin=[])
    ex.run_shard(100)
    n_ev, n_words, data = C.c_uint64(), C.c_uint64(), C.POINTER(C.c_uint64)()
    with pytest.raises(_lib.Sp1HipError, match="family"):
        _lib.check(ex.lib.sp1hip_rv64_precompile_events(ex.h, 15, C.byref(n_ev), C.byref(n_words), C.byref(data)))
    for family in range(15):                                                     # every family answers, empty, with its event size
        _lib.check(ex.lib.sp1hip_rv64_precompile_events(ex.h, family, C.byref(n_ev), C.byref(n_words), C.byref(data)))
        assert n_ev.value == 0 and n_words.value in (24, 28, 34, 40, 44, 58, 64)
