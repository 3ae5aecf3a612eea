"""CPU checks of the Rust side (rust/, source only — this image has no Rust toolchain, so nothing here is compiled):
* rust/sp1-hip-sys/src/lib.rs is exactly what rust/gen_sys.py generates from include/sp1hip.h, and — parsed independently
  of the generator — declares every prototype of the header with the same arity and pointer shape, and every struct with
  the same fields in the same order;
* the exporter's tape (rust/sp1-hip-export/src/recorder.rs, modelled in tests/rust_recorder_model.py) re-records the one
  real machine in the tree, the recursion compress machine, to the same polynomials; the interchange document round-trips
  through sp1_amd.machine; the crates the README names exist and reference only ABI symbols that exist."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HEADER = os.path.join(ROOT, "include", "sp1hip.h")
LIB_RS = os.path.join(ROOT, "rust", "sp1-hip-sys", "src", "lib.rs")


def _strip(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def test_sys_crate_is_what_the_generator_writes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "rust", "gen_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _c_prototypes():
    text = _strip(open(HEADER).read())
    out = {}
    for m in re.finditer(r"\b(sp1hip_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = " ".join(m.group(2).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        # pointer depth per parameter: stars, plus one for an array declarator
        out[m.group(1)] = [p.count("*") + (1 if re.search(r"\[\d*\]$", p) else 0) for p in params]
    return out


def _rust_prototypes():
    text = open(LIB_RS).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (sp1hip_[a-z0-9_]+)\((.*?)\)(?: -> [^;]+)?;", block, flags=re.S):
        args = m.group(2).strip()
        params = [] if not args else [a.strip() for a in re.split(r",\s*(?=[a-z_#0-9]+:)", args)]
        out[m.group(1)] = [len(re.findall(r"\*(?:const|mut)\b", p)) for p in params]
    return out


def test_every_prototype_has_a_matching_declaration():
    c, r = _c_prototypes(), _rust_prototypes()
    assert len(c) >= 90
    assert sorted(c) == sorted(r), (set(c) ^ set(r))
    for name in c:
        assert c[name] == r[name], "%s: pointer shapes differ: C %s, Rust %s" % (name, c[name], r[name])


def test_struct_fields_match():
    text = _strip(open(HEADER).read())
    rust = open(LIB_RS).read()
    n = 0
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(sp1hip_\w+_t)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if decl:
                body = re.sub(r"^(?:const\s+)?\w+\s*\**\s*", "", decl, count=1)
                fields += [re.sub(r"[\*\s]|\[\d+\]", "", f) for f in body.split(",")]
        rname = "Sp1Hip" + "".join(p.capitalize() for p in m.group(2)[len("sp1hip_"):-2].split("_"))
        rm = re.search(r"pub struct %s \{(.*?)\n\}" % rname, rust, flags=re.S)
        assert rm, rname
        assert re.findall(r"pub (\w+):", rm.group(1)) == fields, (rname, fields)
        assert "#[repr(C)]" in rust[max(0, rm.start() - 80):rm.start()]
        n += 1
    assert n >= 12


def test_prover_and_export_crates_use_only_existing_abi_symbols():
    declared = set(_c_prototypes())
    used = set()
    files = 0
    for crate in ("sp1-hip-prover", "sp1-hip-export"):
        src = os.path.join(ROOT, "rust", crate, "src")
        for f in os.listdir(src):
            files += 1
            used |= set(re.findall(r"\bsys::(sp1hip_[a-z0-9_]+)", open(os.path.join(src, f)).read()))
    assert files >= 9 and {"sp1hip_setup", "sp1hip_prove_shard_with_pk", "sp1hip_stage_tables", "sp1hip_pool_submit"} <= used
    assert used <= declared, used - declared
    for f in ("Cargo.toml", "README.md", "sp1-hip-sys/build.rs", "sp1-hip-sys/Cargo.toml", "sp1-hip-prover/Cargo.toml",
              "sp1-hip-export/Cargo.toml"):
        assert os.path.exists(os.path.join(ROOT, "rust", f)), f
    assert "UNCOMPILED" in open(os.path.join(ROOT, "rust", "README.md")).read()


def test_recorder_model_reproduces_the_recursion_machine():
    from rust_recorder_model import record, same_polynomials
    from sp1_amd.air import AirProgram
    from sp1_amd.machine import dump_machine, load_machine
    path = os.path.join(ROOT, "sp1_amd", "machines", "recursion_compress.json")
    doc = json.load(open(path))
    machine = load_machine(doc)
    assert dump_machine(machine) == {k: v for k, v in doc.items() if k != "source"}      # the interchange document round-trips
    rerecorded = []
    for air, inter in machine:
        n_pub = 1 + max([a for op, a, _ in air.instrs if op == 3] + [-1])
        new = AirProgram(air.name, air.main_width, air.prep_width)
        new.instrs = record(air.instrs, air.prep_width, air.main_width, n_pub)
        new.num_constraints = sum(1 for op, _, _ in new.instrs if op == 8)
        rerecorded.append((new, inter))
    assert same_polynomials(machine, rerecorded)
    # and the check does detect a different machine: flip one subtraction
    air0 = rerecorded[0][0]
    k = next(i for i, (op, a, b) in enumerate(air0.instrs) if op == 5 and a != b)
    op, a, b = air0.instrs[k]
    air0.instrs[k] = (op, b, a)
    assert not same_polynomials(machine, rerecorded)
