"""Constraint programs: AIR constraints as data for the zerocheck kernels.

The reference's chips express constraints as Rust generic code over an `AirBuilder`
(`Air::eval(&mut ConstraintSumcheckFolder)`, /root/reference/crates/hypercube/src/folder.rs:L276-L323); a GPU
backend needs them as data, which the reference's own CUDA backend obtains by running `eval` over a
recording builder (/root/reference/sp1-gpu/crates/air/src/ir/bytecode.rs:L27-L110). The program format here is
the same operation set in SSA form, one `[op, a, b]` u32 triple per instruction; instruction k defines
value k:

    0 LOAD_MAIN col | 1 LOAD_PREP col | 2 CONST canonical | 3 PUBLIC idx
    4 ADD a b | 5 SUB a b | 6 MUL a b | 7 NEG a | 8 ASSERT_ZERO a   (the k-th assert gets alpha-power k)
    16 HINT kind col   (optional, no semantics: marks a sub-AIR a prover may evaluate with a fused kernel — the reference's
                        recording builder knows these boundaries too, `eval_external_round` / `eval_internal_rounds` being
                        functions; sp1_amd/csrc/zc_poseidon2.hpp. The oracle and the verifier ignore it.)

`AirProgram` is a tiny builder with operator overloading used by the tests; a JSON dump of the
RISC-V chips from the reference's `crates/core/compiler` maps onto the same triples.
Single-row constraints only (the zerocheck folder exposes no next-row access), degree <= 3.
"""
import numpy as np

LOAD_MAIN, LOAD_PREP, CONST, PUBLIC, ADD, SUB, MUL, NEG, ASSERT_ZERO = range(9)
HINT = 16                 # [16, kind, first main column]: a pseudo-instruction that defines no value and changes no constraint
HINT_POSEIDON2 = 1        # "the next 163 asserts are the Poseidon2 permutation sub-AIR over the 179 columns from that column on"
P = 0x7F000001


class Expr:
    def __init__(self, prog, idx):
        self.prog, self.idx = prog, idx

    def _wrap(self, other):
        return other if isinstance(other, Expr) else self.prog.const(other)

    def __add__(self, o):
        return self.prog._emit(ADD, self.idx, self._wrap(o).idx)

    __radd__ = __add__

    def __sub__(self, o):
        return self.prog._emit(SUB, self.idx, self._wrap(o).idx)

    def __rsub__(self, o):
        return self.prog._emit(SUB, self._wrap(o).idx, self.idx)

    def __mul__(self, o):
        return self.prog._emit(MUL, self.idx, self._wrap(o).idx)

    __rmul__ = __mul__

    def __neg__(self):
        return self.prog._emit(NEG, self.idx, 0)


class AirProgram:
    def __init__(self, name, main_width, prep_width=0, cse=False):
        """cse=True hash-conses the instructions (a repeated load / constant / operation reuses the earlier value,
        ADD and MUL are commutative) — what a recording builder does for the wide chips (Poseidon2: thousands of
        repeated sub-expressions); the value of every constraint is unchanged."""
        self.name, self.main_width, self.prep_width = name, main_width, prep_width
        self.instrs, self.num_constraints = [], 0
        self._seen = {} if cse else None

    def _emit(self, op, a, b):
        if self._seen is not None and op != ASSERT_ZERO:
            key = (op, min(a, b), max(a, b)) if op in (ADD, MUL) else (op, a, b)
            k = self._seen.get(key)
            if k is not None:
                return Expr(self, k)
            self._seen[key] = len(self.instrs)
        self.instrs.append((op, a, b))
        return Expr(self, len(self.instrs) - 1)

    def main(self, col):
        assert 0 <= col < self.main_width
        return self._emit(LOAD_MAIN, col, 0)

    def prep(self, col):
        assert 0 <= col < self.prep_width
        return self._emit(LOAD_PREP, col, 0)

    def const(self, v):
        return self._emit(CONST, int(v) % P, 0)

    def public(self, idx):
        return self._emit(PUBLIC, idx, 0)

    def hint_poseidon2(self, base_col):
        """The 163 asserts that follow are `eval_external_round` (r = 0..7) + `eval_internal_rounds` over main columns
        [base_col, base_col + 179) (hypercube/src/operations/poseidon2/air.rs:L66-L144)."""
        assert 0 <= base_col and base_col + 179 <= self.main_width
        self.instrs.append((HINT, HINT_POSEIDON2, base_col))

    def hint_septic_curve(self, xy_col):
        """The 7 asserts that follow are y^2 - (x^3 + 45 x + 41 z^3) over F_p^7, x = main columns [xy_col, +7), y = the next 7
        (operations/global_interaction.rs:L203-L208)."""
        self.instrs.append((HINT, 2, xy_col))

    def hint_septic_sum(self, xy_col, acc_col, is_real_col):
        """The 14 asserts that follow are sum_checker_x and is_real * sum_checker_y for p1 = main columns [acc_col, +14),
        p2 = [xy_col, +14), p3 = [acc_col + 14, +14) (operations/global_accumulation.rs:L83-L131)."""
        assert xy_col < (1 << 16) and acc_col < (1 << 16) and is_real_col < (1 << 24)
        self.instrs.append((HINT, 3 | (is_real_col << 8), xy_col | (acc_col << 16)))

    def hint_keccak(self, base_col):
        """The 2,858 asserts that follow are one Keccak-f round over main columns [base_col, base_col + 2633) (`KeccakCols`) with the
        round index at base_col + 2638 and is_real at base_col + 2639 (keccak256/air.rs:L40-L164, everything after assert_bool(is_real))."""
        assert 0 <= base_col and base_col + 2640 <= self.main_width
        self.instrs.append((HINT, 5, base_col))

    def hint_mul(self, mul_col, op_b_col):
        """The 16 asserts that follow are MulOperation's product constraints (operations/mul.rs:L196-L236): is_real * (product[k] -
        (m[k] + carry[k - 1] - 256 carry[k])) for the `MulOperation` struct at main columns [mul_col, +45), the chip's five opcode
        flags right behind it (is_real = their sum), the 16-bit limbs of b at [op_b_col, +4) and those of c at [op_b_col + 7, +4)."""
        assert 0 <= mul_col and mul_col + 50 <= self.main_width and 0 <= op_b_col and op_b_col + 11 <= self.main_width
        self.instrs.append((HINT, 6 | (op_b_col << 8), mul_col))

    def hint_polynomial_identity(self, products, rest):
        """The len(rest) asserts that follow are the coefficients of sum_t F_t0(x) F_t1(x) [F_t2(x)] + R(x): assert k is the sum over
        t of the k-th coefficient of the product of term t's two or three polynomials, plus rest[k] — `FieldOpCols`' vanishing
        polynomial (operations/field/util_air.rs:L6-L27), whose 63 coefficients are ~2,000 byte products; a one-coefficient factor
        is a selector (`eval_variable`'s is_add / is_sub / is_mul, field_op.rs:L367-L401). products: [(F0, F1) | (F0, F1, F2)] lists
        of values (Expr), rest: values; every one of them affine in the main columns (a prover checks that, and the identity on a
        pseudo-random row, before it believes the hint). With consecutive alpha powers on consecutive coefficients the batched sum is
        w_0 (sum_t prod_f F_tf(1/alpha) + R(1/alpha)): affine forms over the row instead of convolutions (sp1_amd/csrc/zc_poly.hpp).
        Pseudo-instructions: [16, 7 | terms << 8, len(rest)], then [16, 8 | code << 8, value] with code = 3 t + f for the
        coefficients of factor f of term t, 255 for the rest's, lowest coefficient first."""
        assert len(products) < 80 and all(2 <= len(fs) <= 3 and sum(len(f) - 1 for f in fs) + 1 <= len(rest) for fs in products)
        self.instrs.append((HINT, 7 | (len(products) << 8), len(rest)))
        for t, fs in enumerate(products):
            for f, poly in enumerate(fs):
                for e in poly:
                    self.instrs.append((HINT, 8 | ((3 * t + f) << 8), e.idx))
        for e in rest:
            self.instrs.append((HINT, 8 | (255 << 8), e.idx))

    def assert_zero(self, e):
        self._emit(ASSERT_ZERO, e.idx, 0)
        self.num_constraints += 1

    def assert_eq(self, a, b):
        self.assert_zero(a - b)

    def to_array(self):
        # memoised per instruction count: a prover builds the same program for every shard
        n = len(self.instrs)
        cached = getattr(self, "_array_cache", None)
        if cached is None or cached[0] != n:
            cached = (n, np.array(self.instrs, dtype=np.uint32).reshape(-1, 3))
            self._array_cache = cached
        return cached[1]

    def max_live_registers(self):
        """Registers needed after last-use allocation (what the interpreter kernels size their file by)."""
        return allocate_registers(self.to_array())[1]


def allocate_registers(prog):
    """Linear-scan register allocation for the SSA program. Returns (allocated program, n_registers): an
    [n, 4] array of [op, dst, a, b] where a/b are register numbers (ASSERT_ZERO has no dst)."""
    n = len(prog)
    last_use = [-1] * n
    for k, (op, a, b) in enumerate(prog.tolist()):
        if op in (ADD, SUB, MUL):
            last_use[a] = k
            last_use[b] = k
        elif op in (NEG, ASSERT_ZERO):
            last_use[a] = k
    free, reg_of, out, n_regs = [], {}, [], 0
    for k, (op, a, b) in enumerate(prog.tolist()):
        ra = reg_of.get(a, 0) if op in (ADD, SUB, MUL, NEG, ASSERT_ZERO) else a
        rb = reg_of.get(b, 0) if op in (ADD, SUB, MUL) else b
        # operands whose last use is this instruction free their register before dst is chosen
        if op in (ADD, SUB, MUL, NEG, ASSERT_ZERO):
            for v in {a, b} if op in (ADD, SUB, MUL) else {a}:
                if last_use[v] == k and v in reg_of:
                    free.append(reg_of.pop(v))
        if op == ASSERT_ZERO:
            out.append((op, 0, ra, 0))
            continue
        if last_use[k] < 0:            # dead value: still needs a slot for the write
            dst = free[-1] if free else n_regs
            if not free:
                n_regs += 1
        elif free:
            dst = free.pop()
            reg_of[k] = dst
        else:
            dst = n_regs
            n_regs += 1
            reg_of[k] = dst
        out.append((op, dst, ra, rb))
    return np.array(out, dtype=np.uint32).reshape(-1, 4), max(n_regs, 1)


# ---------------------------------------------------------------------------------------------------------
# Interaction programs: the sends / receives of a chip as data for LogUp-GKR.
#
# The reference stores them as `Interaction { values: Vec<VirtualPairCol>, multiplicity: VirtualPairCol, kind }`
# (/root/reference/crates/hypercube/src/lookup/interaction.rs:L11-L24), a `VirtualPairCol` being
# `sum_k weight_k * column_k + constant` over the preprocessed / main columns of the row. Encoding (u32 words,
# the layout `sp1hip_gkr_chip_t.interactions` expects):
#
#   [n_interactions, then per interaction: is_send, kind, n_values, vcol(multiplicity), vcol(value_0), ...]
#   vcol = [n_terms, constant (canonical), then n_terms x (is_main, column, weight (canonical))]
#
# Order inside a chip: all sends, then all receives (cpu.rs:L86-L92).
class VCol:
    """sum weight * column + constant; columns are ("main", i) or ("prep", i)."""

    def __init__(self, terms=(), constant=0):
        self.terms = [(kind, int(col), int(w) % P) for kind, col, w in terms]
        self.constant = int(constant) % P

    @staticmethod
    def main(col, weight=1):
        return VCol([("main", col, weight)])

    @staticmethod
    def prep(col, weight=1):
        return VCol([("prep", col, weight)])

    @staticmethod
    def const(c):
        return VCol([], c)

    def __add__(self, o):
        o = o if isinstance(o, VCol) else VCol([], o)
        return VCol(self.terms + o.terms, self.constant + o.constant)

    __radd__ = __add__

    def __mul__(self, k):
        return VCol([(a, b, w * int(k)) for a, b, w in self.terms], self.constant * int(k))

    __rmul__ = __mul__

    def words(self):
        out = [len(self.terms), self.constant]
        for kind, col, w in self.terms:
            out += [1 if kind == "main" else 0, col, w]
        return out

    def apply(self, prep_row, main_row):
        acc = self.constant
        for kind, col, w in self.terms:
            acc += w * int((main_row if kind == "main" else prep_row)[col])
        return acc % P


class InteractionProgram:
    """The interactions of one chip (name order across chips is the caller's job: BTreeSet<Chip>)."""

    def __init__(self, name, main_width, prep_width=0):
        self.name, self.main_width, self.prep_width = name, main_width, prep_width
        self.sends, self.receives = [], []

    def send(self, kind, values, multiplicity):
        self.sends.append((int(kind), list(values), multiplicity))

    def receive(self, kind, values, multiplicity):
        self.receives.append((int(kind), list(values), multiplicity))

    @property
    def num_interactions(self):
        return len(self.sends) + len(self.receives)

    def to_array(self):
        key = (len(self.sends), len(self.receives))      # memoised: the same description is sent with every shard
        cached = getattr(self, "_array_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        out = [self.num_interactions]
        for is_send, lst in ((1, self.sends), (0, self.receives)):
            for kind, values, mult in lst:
                out += [is_send, kind, len(values)] + mult.words()
                for v in values:
                    out += v.words()
        arr = np.array(out, dtype=np.uint32)
        self._array_cache = (key, arr)
        return arr
