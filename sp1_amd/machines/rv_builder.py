"""A recording `AirBuilder` for hand transcriptions of the reference's RISC-V chips.

The reference's chips are Rust `Air::eval` bodies generic over an `AirBuilder`; its own GPU backend obtains them as data by
running `eval` over a recording builder (/root/reference/sp1-gpu/crates/air/src/ir). With no Rust toolchain in this image
the bodies are transcribed by hand (`riscv.py`) against THIS builder, which mirrors the builder surface the chips use:

  * p3-air's `AirBuilder` conventions (pinned, for the recursion machine, by the reference's own proof — recursion.py):
    `assert_eq(x, y) = assert_zero(x - y)`, `assert_bool(x) = assert_zero(x (x - 1))`, `assert_one(x) = assert_zero(x - 1)`,
    `when(c).assert_zero(x) = assert_zero(c x)` (nested filters multiply from the inside out),
    `when_not(c) = when_ne(c, 1) = when(c - 1)`;
  * the interaction builders of /root/reference/crates/hypercube/src/air/builder.rs:L44-L260 (`send_byte`, `send_state`, ...)
    and /root/reference/crates/core/machine/src/air/{memory,program,word}.rs, whose message layouts are restated in riscv.py.

A `Sym` is a node of an expression DAG over the columns of ONE row (single-row constraints only: the zerocheck folder has no
next-row access). It carries its affine form while it has one, because interactions store `VirtualPairCol`s — affine
combinations of columns — and the reference's `InteractionBuilder` panics on anything else
(/root/reference/crates/hypercube/src/lookup/builder.rs). Constraints are emitted lazily into an `AirProgram` (hash-consed),
so expressions that only feed interactions cost nothing in the constraint program.
"""
from ..air import AirProgram, InteractionProgram, P, VCol

_CONST, _MAIN, _PREP, _PUB, _ADD, _SUB, _MUL, _NEG = range(8)


class Sym:
    __slots__ = ("b", "op", "x", "y", "lin", "_e")

    def __init__(self, b, op, x, y, lin):
        self.b, self.op, self.x, self.y, self.lin, self._e = b, op, x, y, lin, None

    # -- affine bookkeeping: lin = ({("main"|"prep", col): coeff}, constant) or None
    @property
    def is_const(self):
        return self.lin is not None and not self.lin[0]

    @property
    def const_value(self):
        return self.lin[1]

    def _w(self, o):
        return o if isinstance(o, Sym) else self.b.const(o)

    def __add__(self, o):
        o = self._w(o)
        if o.is_const and o.const_value == 0:
            return self
        if self.is_const and self.const_value == 0:
            return o
        lin = None
        if self.lin is not None and o.lin is not None:
            t = dict(self.lin[0])
            for k, c in o.lin[0].items():
                t[k] = (t.get(k, 0) + c) % P
            lin = ({k: c for k, c in t.items() if c}, (self.lin[1] + o.lin[1]) % P)
            if not lin[0]:
                return self.b.const(lin[1])
        return Sym(self.b, _ADD, self, o, lin)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._w(o)
        if o.is_const and o.const_value == 0:
            return self
        lin = None
        if self.lin is not None and o.lin is not None:
            t = dict(self.lin[0])
            for k, c in o.lin[0].items():
                t[k] = (t.get(k, 0) - c) % P
            lin = ({k: c for k, c in t.items() if c}, (self.lin[1] - o.lin[1]) % P)
            if not lin[0]:
                return self.b.const(lin[1])
        return Sym(self.b, _SUB, self, o, lin)

    def __rsub__(self, o):
        return self._w(o) - self

    def __mul__(self, o):
        o = self._w(o)
        if self.is_const and o.is_const:
            return self.b.const(self.const_value * o.const_value)
        for a, c in ((self, o), (o, self)):
            if c.is_const:
                if c.const_value == 1:
                    return a
                if c.const_value == 0:
                    return self.b.const(0)
                lin = None
                if a.lin is not None:
                    k = c.const_value
                    lin = ({col: (w * k) % P for col, w in a.lin[0].items()}, (a.lin[1] * k) % P)
                return Sym(self.b, _MUL, a, c, lin)
        return Sym(self.b, _MUL, self, o, None)

    __rmul__ = __mul__

    def __neg__(self):
        return self.b.const(0) - self

    # -- lowering
    def expr(self):
        """The AirProgram value of this node (iterative post-order: carry chains are hundreds of nodes deep)."""
        if self._e is not None:
            return self._e
        air = self.b.air
        stack = [self]
        while stack:
            n = stack[-1]
            if n._e is not None:
                stack.pop()
                continue
            if n.op == _CONST:
                n._e = air.const(n.x)
            elif n.op == _MAIN:
                n._e = air.main(n.x)
            elif n.op == _PREP:
                n._e = air.prep(n.x)
            elif n.op == _PUB:
                n._e = air.public(n.x)
            else:
                kids = [k for k in (n.x, n.y) if k is not None and k._e is None]
                if kids:
                    stack.extend(kids)
                    continue
                if n.op == _ADD:
                    n._e = n.x._e + n.y._e
                elif n.op == _SUB:
                    n._e = n.x._e - n.y._e
                elif n.op == _MUL:
                    n._e = n.x._e * n.y._e
            if n._e is not None:
                stack.pop()
        return self._e

    def vcol(self):
        assert self.lin is not None, "interaction value / multiplicity is not affine in the row"
        return VCol([(kind, col, w) for (kind, col), w in self.lin[0].items()], self.lin[1])


class _Filtered:
    """p3-air's FilteredAirBuilder: every asserted expression is multiplied by the condition; sends pass through."""

    def __init__(self, inner, cond):
        self.inner, self.cond = inner, cond

    def assert_zero(self, x):
        self.inner.assert_zero(self.cond * x)

    def __getattr__(self, name):              # everything else is shared with the root builder
        return getattr(self.inner, name)


class _Asserts:
    def _s(self, x):
        return x if isinstance(x, Sym) else self.const(x)

    def assert_eq(self, x, y):
        self.assert_zero(self._s(x) - self._s(y))

    def assert_one(self, x):
        self.assert_zero(self._s(x) - 1)

    def assert_bool(self, x):
        x = self._s(x)
        self.assert_zero(x * (x - 1))

    def when(self, c):
        return _FilteredB(self, self._s(c))

    def when_not(self, c):
        return _FilteredB(self, self._s(c) - 1)

    def assert_all_eq(self, xs, ys):
        xs, ys = list(xs), list(ys)
        assert len(xs) == len(ys)
        for x, y in zip(xs, ys):
            self.assert_eq(x, y)

    def assert_word_eq(self, xs, ys):
        self.assert_all_eq(xs, ys)

    def assert_word_zero(self, xs):
        for x in xs:
            self.assert_zero(self._s(x))

    def if_else(self, c, a, b):
        c = self._s(c)
        return c * self._s(a) + (1 - c) * self._s(b)


class _FilteredB(_Filtered, _Asserts):
    def const(self, v):
        return self.inner.const(v)


class Builder(_Asserts):
    """One chip: `main(i)` / `prep(i)` columns, asserts in call order, sends and receives in call order."""

    def __init__(self, name, main_width, prep_width=0):
        self.name = name
        self.air = AirProgram(name, main_width, prep_width, cse=True)
        self.it = InteractionProgram(name, main_width, prep_width)
        self._consts = {}
        self._cols = {}

    def const(self, v):
        v = int(v) % P
        s = self._consts.get(v)
        if s is None:
            s = self._consts[v] = Sym(self, _CONST, v, None, ({}, v))
        return s

    def main(self, i):
        s = self._cols.get(("main", i))
        if s is None:
            assert 0 <= i < self.air.main_width
            s = self._cols[("main", i)] = Sym(self, _MAIN, i, None, ({("main", i): 1}, 0))
        return s

    def prep(self, i):
        s = self._cols.get(("prep", i))
        if s is None:
            assert 0 <= i < self.air.prep_width
            s = self._cols[("prep", i)] = Sym(self, _PREP, i, None, ({("prep", i): 1}, 0))
        return s

    def public(self, i):
        return Sym(self, _PUB, i, None, None)

    def assert_zero(self, x):
        self.air.assert_zero(self._s(x).expr())

    def send(self, kind, values, mult):
        self.it.send(kind, [self._s(v).vcol() for v in values], self._s(mult).vcol())

    def receive(self, kind, values, mult):
        self.it.receive(kind, [self._s(v).vcol() for v in values], self._s(mult).vcol())


class Cols:
    """Column allocator following a `#[repr(C)]` struct field by field; remembers dotted names for the trace generators."""

    def __init__(self, b, prep=False):
        self.b, self.n, self.prep, self.names = b, 0, prep, {}

    def one(self, name=None):
        c = (self.b.prep if self.prep else self.b.main)(self.n)
        if name is not None:
            self.names[name] = self.n
        self.n += 1
        return c

    def arr(self, k, name=None):
        if name is not None:
            self.names[name] = self.n
        return [self.one() for _ in range(k)]
