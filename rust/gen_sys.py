#!/usr/bin/env python3
"""Generate rust/sp1-hip-sys/src/lib.rs from include/sp1hip.h (no bindgen: the header is plain C89 prototypes, this is a
200-line translation a maintainer can audit). `python rust/gen_sys.py` rewrites the file, `--check` exits 1 if the committed
file is stale. tests/test_rust_glue.py runs the check and, independently, compares names / arities / pointer shapes of
every prototype against the generated declarations."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sp1hip.h")
OUT = os.path.join(ROOT, "rust", "sp1-hip-sys", "src", "lib.rs")

SCALARS = {"int": "c_int", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "float": "f32",
           "double": "f64", "char": "c_char", "void": "c_void"}
HANDLES = {"sp1hip_stream_t": "Stream", "sp1hip_event_t": "Event", "sp1hip_ticket_t": "Ticket", "sp1hip_rv64_vm_t": "Rv64Vm"}


def rust_struct_name(c):
    # sp1hip_fri_config_t -> Sp1HipFriConfig
    body = c[len("sp1hip_"):-2] if c.endswith("_t") else c[len("sp1hip_"):]
    return "Sp1Hip" + "".join(p.capitalize() for p in body.split("_"))


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)


def parse_header(text=None):
    """Returns (enums, structs, opaques, functions). functions: [(name, ret_ctype, [(ctype, pname)], line)]."""
    text = strip_comments(open(HEADER).read() if text is None else text)
    structs, opaques, functions, enums = [], [], [], []
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        items = [(a.strip(), int(b)) for a, b in re.findall(r"(\w+)\s*=\s*(-?\d+)", m.group(1))]
        enums.append((m.group(2), items))
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "uint32_t main_width, prep_width" / "const uint32_t* d_data" / "uint32_t c[4]"
            mm = re.match(r"(.*?)([\w\[\], ]+)$", decl)
            base = decl[:decl.rfind(" ")] if " " in decl else decl
            names = decl[len(base):]
            # split "type a, b": the type is everything up to the first declarator
            first = re.match(r"((?:const\s+)?\w+\s*\**)\s*(.*)$", decl)
            ctype, rest = first.group(1).strip(), first.group(2)
            for nm in rest.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\d+)\]$", nm)
                if arr:
                    fields.append((ctype, arr.group(1), int(arr.group(2))))
                else:
                    stars = nm.count("*")
                    fields.append((ctype + "*" * stars, nm.replace("*", "").strip(), None))
        structs.append((m.group(2), fields))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaques.append(m.group(2))
    proto = re.compile(r"^\s*((?:const\s+)?\w+\s*\**)\s*(sp1hip_\w+)\s*\(([^;{}]*?)\)\s*;", flags=re.M | re.S)
    for m in proto.finditer(text):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        line = text.count("\n", 0, m.start(2)) + 1
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.match(r"(.*?)(\w+)\[\d*\]$", a)
                if arr:                                    # `uint32_t h_commit[8]` decays to a pointer
                    params.append((arr.group(1).strip() + "*", arr.group(2)))
                    continue
                mm = re.match(r"(.*?)(\w+)$", a)
                params.append((mm.group(1).strip(), mm.group(2)))
        functions.append((name, ret, params, line))
    return enums, structs, opaques, functions


def rust_type(ctype, known_structs, opaques):
    """C type (pointer stars attached, e.g. `const uint32_t*`, `uint32_t* const*`) -> Rust FFI type."""
    t = ctype.replace(" ", "")
    # peel pointers from the right: each level is `*` optionally followed by `const`
    levels = []
    while t.endswith("*") or t.endswith("*const"):
        if t.endswith("*const"):
            t = t[:-len("*const")]
            levels.append("const_ptr_level")
        else:
            t = t[:-1]
            levels.append("ptr")
    const_base = t.startswith("const")
    if const_base:
        t = t[len("const"):]
    if t in SCALARS:
        base = SCALARS[t]
    elif t in HANDLES:
        base = HANDLES[t]
    elif t in opaques or t in known_structs:
        base = rust_struct_name(t)
    else:
        raise ValueError("unknown C type %r" % ctype)
    out = base
    # innermost pointer points at the (possibly const) base; a `* const` level is a const pointer to what is to its left
    for i, lv in enumerate(reversed(levels)):
        innermost = i == 0
        is_const = const_base if innermost else (list(reversed(levels))[i - 1] == "const_ptr_level")
        out = ("*const " if is_const else "*mut ") + out
    return out


def generate():
    enums, structs, opaques, functions = parse_header()
    known = [s[0] for s in structs]
    w = []
    w.append("//! `sp1-hip-sys`: raw FFI declarations for `libsp1hip.so` (include/sp1hip.h), the MI355X backend of SP1's")
    w.append("//! core-shard proving path. GENERATED by rust/gen_sys.py from the header — do not edit; the repository's CPU")
    w.append("//! tests fail if this file and the header disagree. UNCOMPILED in the repository that produced it (its image has")
    w.append("//! no Rust toolchain): build it next to the reference with the commands in rust/README.md.")
    w.append("//!")
    w.append("//! Conventions are those of the reference's own GPU FFI crate (sp1-gpu/crates/sys/src/runtime.rs:L3-L172): raw")
    w.append("//! device pointers, field elements as Montgomery `u32` words (`KoalaBear` is `#[repr(transparent)]` over that")
    w.append("//! word), every compute call asynchronous on the given stream, `c_int` status codes, no unwinding across the ABI.")
    w.append("#![allow(non_camel_case_types, clippy::too_many_arguments, clippy::missing_safety_doc)]")
    w.append("")
    w.append("use core::ffi::{c_char, c_int, c_void};")
    w.append("")
    w.append("/// `hipStream_t`; null = the default stream.")
    w.append("pub type Stream = *mut c_void;")
    w.append("/// `hipEvent_t`.")
    w.append("pub type Event = *mut c_void;")
    w.append("/// `sp1hip_ticket_t`: one shard submitted to a prover pool.")
    w.append("pub type Ticket = u64;")
    w.append("/// `sp1hip_rv64_vm_t`: one guest execution (host code).")
    w.append("pub type Rv64Vm = *mut c_void;")
    w.append("")
    for name, items in enums:
        w.append("/// `%s`" % name)
        for item, val in items:
            w.append("pub const %s: c_int = %d;" % (item, val))
        w.append("")
    for o in opaques:
        w.append("/// Opaque `%s`." % o)
        w.append("#[repr(C)]")
        w.append("pub struct %s {" % rust_struct_name(o))
        w.append("    _private: [u8; 0],")
        w.append("}")
        w.append("")
    for name, fields in structs:
        w.append("/// `%s`" % name)
        w.append("#[repr(C)]")
        w.append("#[derive(Clone, Copy, Debug)]")
        w.append("pub struct %s {" % rust_struct_name(name))
        for ctype, fname, arr in fields:
            rt = rust_type(ctype, known, opaques)
            w.append("    pub %s: %s," % (fname, "[%s; %d]" % (rt, arr) if arr else rt))
        w.append("}")
        w.append("")
    w.append('#[link(name = "sp1hip", kind = "dylib")]')
    w.append('extern "C" {')
    for name, ret, params, line in functions:
        args = ", ".join("%s: %s" % (("r#%s" % p if p in ("type", "in", "fn") else p), rust_type(t, known, opaques)) for t, p in params)
        rett = ""
        if ret != "void":
            rett = " -> " + rust_type(ret, known, opaques)
        w.append("    /// include/sp1hip.h:L%d" % line)
        w.append("    pub fn %s(%s)%s;" % (name, args, rett))
    w.append("}")
    w.append("")
    w.append("/// The calling thread's last error message (`sp1hip_last_error`) as an owned string.")
    w.append("pub fn last_error() -> String {")
    w.append("    // SAFETY: the library returns a NUL-terminated thread-local buffer that stays valid until the next failing call")
    w.append("    // on this thread.")
    w.append("    unsafe { core::ffi::CStr::from_ptr(sp1hip_last_error()) }.to_string_lossy().into_owned()")
    w.append("}")
    w.append("")
    w.append("/// `Ok(())` for `SP1HIP_SUCCESS`, otherwise the status with the library's message.")
    w.append("pub fn check(status: c_int) -> Result<(), (c_int, String)> {")
    w.append("    if status == SP1HIP_SUCCESS {")
    w.append("        Ok(())")
    w.append("    } else {")
    w.append("        Err((status, last_error()))")
    w.append("    }")
    w.append("}")
    return "\n".join(w) + "\n"


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("rust/sp1-hip-sys/src/lib.rs is stale: run python rust/gen_sys.py", file=sys.stderr)
            sys.exit(1)
    else:
        with open(OUT, "w") as f:
            f.write(text)
        print("wrote", OUT)
