"""The constraint-program compiler of the zerocheck interpreter, checked on the CPU (no GPU): `sp1hip_zerocheck_plan_eval`
plans a program exactly as the prover does — immediates folded, instructions reordered, registers allocated with operand
forwarding, multiply-adds fused, forwarded operands put first (RSUB), and the chunked / undivided / finely cut forms — and
interprets the result on one row. Every form must give, constraint by constraint, what a direct evaluation of the
caller's SSA program gives (tests/machine_check.py, numpy). Covers the synthetic AIRs of the GPU tests and the eight real
chips of the recursion machine."""
import ctypes as C

import numpy as np
import pytest

from machine_check import P, constraint_values
from sp1_amd import _lib
from sp1_amd.machines import recursion, riscv

zc_airs = pytest.importorskip("zc_airs")

R = (1 << 32) % P


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _plan_eval(lib, air, main_row, prep_row, publics, form):
    prog = np.ascontiguousarray(air.to_array().reshape(-1), dtype=np.uint32)
    n_c = air.num_constraints
    out = np.zeros(max(n_c, 1), dtype=np.uint32)
    stats = np.zeros(3, dtype=np.uint32)
    to_m = lambda v: np.ascontiguousarray((v.astype(np.uint64) * np.uint64(R)) % np.uint64(P), dtype=np.uint32)
    m, p, pub = to_m(main_row), to_m(prep_row), to_m(publics)
    st = lib.sp1hip_zerocheck_plan_eval(_u32p(prog), len(air.instrs), air.main_width, air.prep_width, _u32p(m), _u32p(p), _u32p(pub),
                                        len(pub), form, _u32p(out), n_c, _u32p(stats))
    assert st == 0, lib.sp1hip_last_error().decode()
    r_inv = pow(1 << 32, -1, P)
    return (out[:n_c].astype(np.uint64) * np.uint64(r_inv)) % np.uint64(P), stats


def _check(lib, air, seed, n_publics):
    rng = np.random.default_rng(seed)
    main = rng.integers(0, P, size=(3, max(air.main_width, 1)), dtype=np.uint64)
    prep = rng.integers(0, P, size=(3, max(air.prep_width, 1)), dtype=np.uint64)
    main[2] = 0                                       # the all-zero row (padded-row adjustment) is one of the rows
    prep[2] = 0
    publics = rng.integers(0, P, size=max(n_publics, 1), dtype=np.uint64)
    want = constraint_values(air, prep, main, publics)
    for row in range(3):
        for form in range(4):
            got, stats = _plan_eval(lib, air, main[row], prep[row], publics, form)
            assert np.array_equal(got, want[row]), (air.name, row, form)
            assert stats[0] > 0 or air.num_constraints == 0


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def test_compiled_forms_of_the_synthetic_airs_evaluate_like_the_ssa(lib):
    airs = [zc_airs.air_mul(), zc_airs.air_affine(), zc_airs.air_sbox(), zc_airs.air_chain(), zc_airs.air_chain(300),
            zc_airs.air_manyregs(), zc_airs.air_manyregs(120)]
    for k, air in enumerate(airs):
        _check(lib, air, 100 + k, 8)


def test_compiled_forms_of_the_recursion_chips_evaluate_like_the_ssa(lib):
    for k, (air, _) in enumerate(recursion.compress_machine()):
        if air.num_constraints:
            _check(lib, air, 200 + k, 200)


def test_plan_eval_rejects_a_wrong_constraint_count(lib):
    air = zc_airs.air_mul()
    prog = np.ascontiguousarray(air.to_array().reshape(-1), dtype=np.uint32)
    row = np.zeros(max(air.main_width, 1), dtype=np.uint32)
    out = np.zeros(air.num_constraints + 1, dtype=np.uint32)
    st = lib.sp1hip_zerocheck_plan_eval(_u32p(prog), len(air.instrs), air.main_width, air.prep_width, _u32p(row), _u32p(row), _u32p(row),
                                        1, 0, _u32p(out), air.num_constraints + 1, None)
    assert st != 0


def _random_air(rng, n_ops, main_w, prep_w, n_publics):
    """A random SSA program: loads / constants / publics feeding a random DAG of ADD / SUB / MUL / NEG, asserts sprinkled in;
    biased towards the shapes the compiler rewrites (products by constants feeding sums, re-use of the last value, squares)."""
    from sp1_amd.air import AirProgram
    air = AirProgram("Fuzz", main_w, prep_w)
    vals = []
    for c in range(main_w):
        vals.append(air.main(c))
    for c in range(prep_w):
        vals.append(air.prep(c))
    for _ in range(3):
        vals.append(air.const(int(rng.integers(0, P))))
    if n_publics:
        vals.append(air.public(int(rng.integers(0, n_publics))))
    for _ in range(n_ops):
        kind = rng.integers(0, 10)
        last = vals[-1]
        pick = lambda: vals[int(rng.integers(0, len(vals)))] if rng.integers(0, 3) else last
        if kind <= 2:
            v = pick() + pick()
        elif kind <= 4:
            v = pick() - pick()
        elif kind <= 6:
            v = pick() * pick()
        elif kind == 7:
            v = pick() + pick() * air.const(int(rng.integers(0, P)))       # multiply-add by a constant
        elif kind == 8:
            v = pick() - air.const(int(rng.integers(1, 1 << 16))) * pick()
        else:
            v = -pick()
        vals.append(v)
        if rng.integers(0, 6) == 0:
            air.assert_zero(vals[int(rng.integers(max(0, len(vals) - 8), len(vals)))])
    air.assert_zero(vals[-1])
    return air


@pytest.mark.parametrize("seed", range(24))
def test_compiled_forms_of_random_programs_evaluate_like_the_ssa(lib, seed):
    rng = np.random.default_rng(7000 + seed)
    air = _random_air(rng, int(rng.integers(5, 500)), int(rng.integers(1, 40)), int(rng.integers(0, 6)), 4)
    _check(lib, air, 9000 + seed, 4)


def test_compiled_forms_of_the_riscv_chips_evaluate_like_the_ssa(lib):
    """Incl. Global, whose Poseidon2 block is evaluated by the fused pieces (zc_poseidon2.hpp) in forms 1-3: the host model of
    those pieces must give the 163 hinted constraints exactly what the SSA gives."""
    for k, name in enumerate(("Global", "Mul", "ShiftRight", "Branch", "LoadByte", "StoreByte", "Addi")):
        _check(lib, riscv.chip(name)[0], 300 + k, 4)


def test_compiled_forms_of_the_round_5_chips_evaluate_like_the_ssa(lib):
    """DivRem, the syscall chips, the Keccak controller — and KeccakPermute, whose 2,858 hinted constraints are evaluated by the
    fused Keccak pieces (zc_keccak.hpp) in forms 1-3: their host model must give every constraint what the SSA gives (random rows,
    the all-zero row included; SyscallInstrs reads 148 public values)."""
    for k, name in enumerate(("KeccakPermute", "DivRem", "SyscallInstrs", "SyscallCore", "KeccakPermuteControl", "MemoryGlobalInit")):
        _check(lib, riscv.chip(name)[0], 400 + k, 160)


def test_compiled_forms_of_the_precompile_chips_evaluate_like_the_ssa(lib):
    """The precompile chips of the real-program shards: the curve chips are the longest programs the interpreter runs (23k
    instructions)."""
    names = ("Secp256k1AddAssign", "Secp256k1DoubleAssign", "Uint256MulMod", "ShaCompress", "ShaExtend", "Poseidon2",
             # the other curve / tower chips (48-limb bls12-381 operands: up to 71k instructions; inner products and the
             # operation-selected polynomials of the Fp chips)
             "Secp256r1AddAssign", "Bn254DoubleAssign", "Bls12381AddAssign", "Bls12381DoubleAssign", "Bn254FpOpAssign", "Bls12381FpOpAssign",
             "Bn254Fp2AddSubAssign", "Bls12381Fp2AddSubAssign", "Bn254Fp2MulAssign", "Bls12381Fp2MulAssign", "EdAddAssign", "EdDecompress", "Uint256Ops")
    for k, name in enumerate(names):
        air = riscv.chip(name)[0]
        _check(lib, air, 500 + k, 8)
        rng = np.random.default_rng(k)
        row = rng.integers(0, P, size=air.main_width, dtype=np.uint64)
        _, stats = _plan_eval(lib, air, row, np.zeros(1, dtype=np.uint64), np.zeros(8, dtype=np.uint64), 1)
        assert stats[2] * 16 * 64 <= 160 * 1024, (name, stats)          # one wave's extension registers fit the CU's LDS
        if name in ("Secp256k1AddAssign", "Secp256k1DoubleAssign", "Uint256MulMod"):
            # the FieldOpCols chips take the rematerialising schedule (columns re-loaded at every use): 214 / 225 / 195 registers
            # with every column kept -> at most 20, for a chunked program under three times the length of the SSA
            assert stats[2] <= 20 and stats[0] < 3 * len(air.instrs), (name, stats)


def test_planner_rejects_overlapping_hints(lib):
    """each hint is checked against the SSA on its own; two hints over the same constraints / columns (a duplicated
    HINT) would each pass and then be counted twice. The planner must refuse the program."""
    import copy
    from sp1_amd.air import HINT
    air = riscv.chip("Global")[0]
    hints = [k for k, ins in enumerate(air.instrs) if ins[0] == HINT]
    assert hints, "the Global chip carries hints"
    dup = copy.copy(air)
    dup.instrs = list(air.instrs)
    dup.instrs.insert(hints[0] + 1, air.instrs[hints[0]])          # the same hint twice in a row
    # the duplicated pseudo-instruction shifts the value numbering: renumber operand references behind it
    at = hints[0] + 1
    fixed = []
    from sp1_amd import air as A
    for k, ins in enumerate(dup.instrs):
        op, a, b = ins
        if k > at and op not in (A.CONST, A.LOAD_MAIN, A.LOAD_PREP, A.PUBLIC, HINT):
            a = a + 1 if a >= at else a
            if op in (A.ADD, A.SUB, A.MUL):
                b = b + 1 if b >= at else b
        fixed.append((op, a, b))
    dup.instrs = fixed
    rng = np.random.default_rng(1)
    row = rng.integers(0, P, size=air.main_width, dtype=np.uint64)
    prow = rng.integers(0, P, size=max(air.prep_width, 1), dtype=np.uint64)
    pub = rng.integers(0, P, size=4, dtype=np.uint64)
    prog = np.ascontiguousarray(dup.to_array().reshape(-1), dtype=np.uint32)
    out = np.zeros(air.num_constraints, dtype=np.uint32)
    stats = np.zeros(3, dtype=np.uint32)
    u = lambda v: np.ascontiguousarray(v, dtype=np.uint32)
    st = lib.sp1hip_zerocheck_plan_eval(_u32p(prog), len(dup.instrs), air.main_width, air.prep_width, _u32p(u(row)), _u32p(u(prow)),
                                        _u32p(u(pub)), 4, 1, _u32p(out), air.num_constraints, _u32p(stats))
    assert st != 0 and b"same constraints" in lib.sp1hip_last_error()


def test_polynomial_identity_tables_collapse_to_the_batched_constraints(lib):
    """Hint kind 7 (zc_poly.hpp): FieldOpCols' coefficient constraints carry consecutive alpha powers, so the kernels evaluate
    w_0 (sum of products of affine forms at 1 / alpha + the rest) instead of the convolution. sp1hip_zerocheck_poly_check builds
    the device tables for a given alpha as the prover does and evaluates them on one row like the kernels: the result must equal
    the sum of alpha^(n - 1 - k) C_k(row) over the covered constraints — for every chip that carries the hint (two- and
    three-factor terms, selectors, a modulus from memory, 32- and 48-limb fields), random rows, the zero row, several alphas."""
    from sp1_amd._lib import Ext
    names = ("Secp256k1AddAssign", "Secp256k1DoubleAssign", "Uint256MulMod", "Bn254FpOpAssign", "Bls12381FpOpAssign", "Bn254Fp2AddSubAssign",
             "Bls12381Fp2MulAssign", "EdAddAssign", "EdDecompress", "Uint256Ops", "Bls12381AddAssign")
    rng = np.random.default_rng(77)
    for name in names:
        air = riscv.chip(name)[0]
        prog = np.ascontiguousarray(air.to_array().reshape(-1), dtype=np.uint32)
        for trial in range(3):
            row = rng.integers(0, P, size=air.main_width, dtype=np.uint64) if trial < 2 else np.zeros(air.main_width, dtype=np.uint64)
            m = np.ascontiguousarray((row * np.uint64(R)) % np.uint64(P), dtype=np.uint32)
            alpha = Ext()
            for k in range(4):
                alpha.c[k] = int(rng.integers(1, P))
            a, b, n = Ext(), Ext(), C.c_uint32()
            st = lib.sp1hip_zerocheck_poly_check(_u32p(prog), len(air.instrs), air.main_width, air.prep_width, _u32p(m), alpha, C.byref(a), C.byref(b), C.byref(n))
            assert st == 0, lib.sp1hip_last_error().decode()
            assert n.value >= 1, name
            assert list(a.c) == list(b.c), (name, trial)
    # a chip without the hint: nothing to check, both sums empty
    air = riscv.chip("Add")[0]
    prog = np.ascontiguousarray(air.to_array().reshape(-1), dtype=np.uint32)
    a, b, n = Ext(), Ext(), C.c_uint32()
    alpha = Ext()
    alpha.c[0] = 5
    assert lib.sp1hip_zerocheck_poly_check(_u32p(prog), len(air.instrs), air.main_width, air.prep_width, _u32p(np.zeros(air.main_width, dtype=np.uint32)), alpha,
                                           C.byref(a), C.byref(b), C.byref(n)) == 0 and n.value == 0
