"""Machine descriptions: the chips of a STARK machine as DATA (the "constraint export" seam).

SP1's chips are Rust generic code: constraints are `Air::eval(&mut impl AirBuilder)`, lookups are
`chip.sends()` / `chip.receives()` (/root/reference/crates/hypercube/src/{air,lookup}). A GPU backend needs both as
data; the reference's own CUDA backend gets them by running `eval` over a recording builder
(/root/reference/sp1-gpu/crates/air/src/ir/bytecode.rs:L27-L110). This module fixes the interchange format between
that Rust-side export (a ~100-line recording `AirBuilder`, sketched in INTEGRATION.md §6; it cannot be built in
this image, which has no Rust toolchain) and this library: one JSON document per machine.

    {
      "field": "KoalaBear",
      "chips": [
        {
          "name": "Add",                        # chips sorted by name (BTreeSet<Chip> order)
          "main_width": 17, "preprocessed_width": 0,
          "constraints": [[op, a, b], ...],     # SSA program of sp1_amd.air: instruction k defines value k
                                                #   0 LOAD_MAIN col | 1 LOAD_PREP col | 2 CONST canonical | 3 PUBLIC idx
                                                #   4 ADD a b | 5 SUB a b | 6 MUL a b | 7 NEG a | 8 ASSERT_ZERO a
          "sends":    [{"kind": 5, "multiplicity": VCOL, "values": [VCOL, ...]}, ...],
          "receives": [ ... same shape ... ]
        }, ...
      ]
    }
    VCOL = {"constant": c, "terms": [["main" | "prep", column, weight], ...]}    (canonical field elements)

`load_machine` turns it into (AirProgram, InteractionProgram) pairs — what `api.prove_shard`, `api.zerocheck` and
`api.logup_gkr` take next to the device traces — after validating every index; `dump_machine` is its inverse.
"""
import json

from .air import ADD, ASSERT_ZERO, CONST, HINT, LOAD_MAIN, LOAD_PREP, MUL, NEG, PUBLIC, SUB, AirProgram, InteractionProgram, P, VCol

_BINARY, _UNARY = (ADD, SUB, MUL), (NEG, ASSERT_ZERO)


def _vcol_from(obj, main_width, prep_width, where):
    terms = []
    for kind, col, weight in obj.get("terms", []):
        if kind not in ("main", "prep"):
            raise ValueError("%s: column kind %r" % (where, kind))
        if not 0 <= int(col) < (main_width if kind == "main" else prep_width):
            raise ValueError("%s: %s column %d out of range" % (where, kind, col))
        if not 0 <= int(weight) < P:
            raise ValueError("%s: weight not a canonical field element" % where)
        terms.append((kind, int(col), int(weight)))
    c = int(obj.get("constant", 0))
    if not 0 <= c < P:
        raise ValueError("%s: constant not a canonical field element" % where)
    return VCol(terms, c)


def _vcol_to(v):
    return {"constant": v.constant, "terms": [[k, c, w] for k, c, w in v.terms]}


def load_machine(doc):
    """doc: a JSON string, a file object or the parsed dict. Returns [(AirProgram, InteractionProgram)] in name order."""
    if hasattr(doc, "read"):
        doc = json.load(doc)
    elif isinstance(doc, (str, bytes)):
        doc = json.loads(doc)
    if doc.get("field", "KoalaBear") != "KoalaBear":
        raise ValueError("only KoalaBear machines are supported")
    out, prev = [], None
    for chip in doc["chips"]:
        name, mw, pw = chip["name"], int(chip["main_width"]), int(chip.get("preprocessed_width", 0))
        if prev is not None and not prev < name:
            raise ValueError("chips must be sorted by name and distinct (%r after %r)" % (name, prev))
        prev = name
        air = AirProgram(name, mw, pw)
        for k, (op, a, b) in enumerate(chip["constraints"]):
            op, a, b = int(op), int(a), int(b)
            where = "%s constraint instruction %d" % (name, k)
            if op == LOAD_MAIN and not 0 <= a < mw or op == LOAD_PREP and not 0 <= a < pw:
                raise ValueError("%s: column %d out of range" % (where, a))
            if op == CONST and not 0 <= a < P:
                raise ValueError("%s: constant not canonical" % where)
            if op in _BINARY and not (0 <= a < k and 0 <= b < k) or op in _UNARY and not 0 <= a < k:
                raise ValueError("%s: operand is not an earlier value" % where)
            if not (0 <= op <= ASSERT_ZERO or op == HINT) or op == PUBLIC and a < 0:
                raise ValueError("%s: bad opcode / operand" % where)
            air.instrs.append((op, a, b))
            air.num_constraints += op == ASSERT_ZERO
        inter = InteractionProgram(name, mw, pw)
        for key, add in (("sends", inter.send), ("receives", inter.receive)):
            for j, it in enumerate(chip.get(key, [])):
                where = "%s %s[%d]" % (name, key, j)
                add(int(it["kind"]), [_vcol_from(v, mw, pw, where) for v in it["values"]], _vcol_from(it["multiplicity"], mw, pw, where))
        out.append((air, inter))
    return out


def dump_machine(chips):
    """chips: [(AirProgram, InteractionProgram)] -> the JSON-able dict of the format above (sorted by name)."""
    doc = {"field": "KoalaBear", "chips": []}
    for air, inter in sorted(chips, key=lambda c: c[1].name):
        assert (air.main_width, air.prep_width) == (inter.main_width, inter.prep_width)
        entry = {"name": inter.name, "main_width": air.main_width, "preprocessed_width": air.prep_width,
                 "constraints": [[int(op), int(a), int(b)] for op, a, b in air.instrs]}
        for key, lst in (("sends", inter.sends), ("receives", inter.receives)):
            entry[key] = [{"kind": kind, "multiplicity": _vcol_to(mult), "values": [_vcol_to(v) for v in values]}
                          for kind, values, mult in lst]
        doc["chips"].append(entry)
    return doc
