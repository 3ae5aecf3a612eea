"""A Python model of rust/sp1-hip-export/src/recorder.rs (the recording AirBuilder's tape) and the acceptance check of
the exporter: two machine descriptions are THE SAME MACHINE if, on random rows, every constraint has the same value and
every interaction the same (kind, multiplicity, values) — instruction numbering is free."""
import numpy as np

from sp1_amd.air import ADD, ASSERT_ZERO, CONST, HINT, LOAD_MAIN, LOAD_PREP, MUL, NEG, PUBLIC, SUB

P = 0x7F000001


class Tape:
    """`Tape::emit` of recorder.rs: hash-consing on (op, a, b) with ADD / MUL commutative; asserts are never merged."""

    def __init__(self):
        self.instrs, self.seen = [], {}

    def emit(self, op, a, b=0):
        if op == ASSERT_ZERO:
            self.instrs.append((op, a, b))
            return len(self.instrs) - 1
        key = (op, min(a, b), max(a, b)) if op in (ADD, MUL) else (op, a, b)
        k = self.seen.get(key)
        if k is not None:
            return k
        self.instrs.append((op, a, b))
        self.seen[key] = len(self.instrs) - 1
        return len(self.instrs) - 1


def record(program, prep_width, main_width, n_publics):
    """`record()` of recorder.rs driven by an existing SSA program standing for `air.eval`: columns and public values are
    loaded first (in order), then every operation of `program` is re-emitted through the tape."""
    t = Tape()
    prep = [t.emit(LOAD_PREP, c) for c in range(prep_width)]
    main = [t.emit(LOAD_MAIN, c) for c in range(main_width)]
    pub = [t.emit(PUBLIC, i) for i in range(n_publics)]
    new = {}
    for k, (op, a, b) in enumerate(program):
        if op == LOAD_MAIN:
            new[k] = main[a]
        elif op == LOAD_PREP:
            new[k] = prep[a]
        elif op == PUBLIC:
            new[k] = pub[a] if a < n_publics else t.emit(PUBLIC, a)
        elif op == CONST:
            new[k] = t.emit(CONST, a)
        elif op in (ADD, SUB, MUL):
            new[k] = t.emit(op, new[a], new[b])
        elif op == NEG:
            new[k] = t.emit(NEG, new[a])
        elif op == HINT:                                  # `Tape::hint`: copied through, defines no value
            t.instrs.append((op, a, b))
        else:
            new[k] = t.emit(ASSERT_ZERO, new[a])
    return t.instrs


def constraint_values(instrs, prep_row, main_row, publics):
    vals, out = [], []
    for op, a, b in instrs:
        if op == LOAD_MAIN:
            v = int(main_row[a])
        elif op == LOAD_PREP:
            v = int(prep_row[a])
        elif op == CONST:
            v = a
        elif op == PUBLIC:
            v = int(publics[a])
        elif op == ADD:
            v = (vals[a] + vals[b]) % P
        elif op == SUB:
            v = (vals[a] - vals[b]) % P
        elif op == MUL:
            v = vals[a] * vals[b] % P
        elif op == NEG:
            v = -vals[a] % P
        elif op == HINT:
            v = 0
        else:
            v = vals[a]
            out.append(v)
        vals.append(v)
    return out


def same_polynomials(machine_a, machine_b, rows=3, seed=11, n_publics=256):
    """machine_*: [(AirProgram, InteractionProgram)] from sp1_amd.machine.load_machine."""
    rng = np.random.default_rng(seed)
    if [i.name for _, i in machine_a] != [i.name for _, i in machine_b]:
        return False
    for (air_a, int_a), (air_b, int_b) in zip(machine_a, machine_b):
        if (air_a.main_width, air_a.prep_width, air_a.num_constraints) != (air_b.main_width, air_b.prep_width, air_b.num_constraints):
            return False
        for _ in range(rows):
            main = rng.integers(0, P, air_a.main_width)
            prep = rng.integers(0, P, air_a.prep_width)
            pub = rng.integers(0, P, n_publics)
            if constraint_values(air_a.instrs, prep, main, pub) != constraint_values(air_b.instrs, prep, main, pub):
                return False
            for la, lb in ((int_a.sends, int_b.sends), (int_a.receives, int_b.receives)):
                if len(la) != len(lb):
                    return False
                for (ka, va, ma), (kb_, vb, mb) in zip(la, lb):
                    if ka != kb_ or ma.apply(prep, main) != mb.apply(prep, main) or len(va) != len(vb):
                        return False
                    if any(x.apply(prep, main) != y.apply(prep, main) for x, y in zip(va, vb)):
                        return False
    return True
