"""CPU-side checks of the drop-in boundary: libsp1hip.so builds for gfx950, loads without a GPU, exports
every function include/sp1hip.h declares, validates arguments before touching a device, and its
host-side transcript (no kernels involved) matches the oracle."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import pyoracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_hip()
    from sp1_amd import _lib
    return _lib.load()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "sp1hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sp1hip_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    from sp1_amd import _lib
    names = _declared()
    assert len(names) >= 50
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "missing export: " + n
    assert sorted(n for n, _, _ in _lib.PROTOTYPES) == names, "ctypes prototypes out of sync with the header"


def test_header_is_plain_c():
    import subprocess
    src = '#include "sp1hip.h"\nint main(void){ sp1hip_fri_config_t c = {2,124,16}; return c.log_blowup - 2; }\n'
    p = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I",
                        os.path.join(ROOT, "include"), "-x", "c", "-"], input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()


def test_argument_validation_needs_no_device(lib):
    assert lib.sp1hip_version().startswith(b"sp1hip")
    n = C.c_int(-1)
    assert lib.sp1hip_device_count(C.byref(n)) == 0 and n.value >= 0
    assert lib.sp1hip_rs_encode_batch(None, None, 20, 5, 1, None) == -1
    assert b"two-adicity" in lib.sp1hip_last_error()
    assert lib.sp1hip_rs_encode_batch(None, None, 3, 1, 0, None) == 0          # empty batch is a no-op
    assert lib.sp1hip_merkle_commit(None, 0, 3, None, None, None) == -1
    from sp1_amd._lib import Ext
    assert lib.sp1hip_fold_even_odd(None, 0, Ext(), None, None) == -1
    assert lib.sp1hip_challenger_observe(None, None, 0) == -1
    # the prover-level entry points reject malformed input before touching a device
    from sp1_amd._lib import FriConfig, GkrChip, ShardParams
    n = C.c_size_t(0)
    assert lib.sp1hip_jagged_prove(None, 3, None, 0, None, None, FriConfig(2, 124, 16), None, None, C.byref(n), None) == -1
    assert lib.sp1hip_logup_gkr_prove(None, 0, 3, None, None, C.byref(n), None) == -1
    assert lib.sp1hip_prove_shard(None, 0, None, 0, None, ShardParams(3, 2, 2, FriConfig(2, 124, 16)), None, None, C.byref(n), None) == -1
    h = C.c_void_p()
    assert lib.sp1hip_challenger_new(C.byref(h)) == 0
    words = np.array([1, 1, 5, 1, 0, 0, 1, 0, 7, 1], np.uint32)            # a value column index 7 in a 2-column chip
    bad = (GkrChip * 1)(GkrChip(b"X", words.ctypes.data_as(C.POINTER(C.c_uint32)), words.size, 2, 0, None, None, 0))
    assert lib.sp1hip_logup_gkr_prove(bad, 1, 3, h, None, C.byref(n), None) == -1
    assert b"value column" in lib.sp1hip_last_error() or b"multiplicity" in lib.sp1hip_last_error()
    none = np.array([0], np.uint32)                                      # a chip without interactions
    two = (GkrChip * 2)(GkrChip(b"B", none.ctypes.data_as(C.POINTER(C.c_uint32)), 1, 2, 0, None, None, 0),
                        GkrChip(b"A", none.ctypes.data_as(C.POINTER(C.c_uint32)), 1, 2, 0, None, None, 0))
    assert lib.sp1hip_logup_gkr_prove(two, 2, 3, h, None, C.byref(n), None) == -1 and b"sorted" in lib.sp1hip_last_error()
    lib.sp1hip_challenger_free(h)
    # staging: nothing to do is fine, missing buffers and over-wide tables are rejected up front
    from sp1_amd._lib import HostTable
    assert lib.sp1hip_stage_tables(None, 0, None, None) == 0
    outs = (C.c_void_p * 1)(None)
    assert lib.sp1hip_stage_tables((HostTable * 1)(HostTable(None, 0, 7)), 1, outs, None) == 0     # an empty trace
    assert lib.sp1hip_stage_tables((HostTable * 1)(HostTable(None, 4, 7)), 1, outs, None) == -1
    assert b"null table data" in lib.sp1hip_last_error()
    assert lib.sp1hip_stage_tables((HostTable * 1)(HostTable(words.ctypes.data, 1, 1 << 20)), 1, outs, None) == -1
    assert lib.sp1hip_host_register(None, 16) == -1 and lib.sp1hip_host_unregister(None) == -1


def test_host_transcript_matches_oracle(lib):
    from sp1_amd._lib import Ext
    h = C.c_void_p()
    assert lib.sp1hip_challenger_new(C.byref(h)) == 0
    o = orc.Challenger()
    rng = np.random.default_rng(4)
    for step in range(300):
        op = rng.integers(0, 4)
        if op == 0:
            xs = orc.random_felts((int(rng.integers(1, 20)),), step)
            assert lib.sp1hip_challenger_observe(h, xs.ctypes.data_as(C.POINTER(C.c_uint32)), xs.size) == 0
            o.observe(xs)
        elif op == 1:
            v = C.c_uint32()
            assert lib.sp1hip_challenger_sample(h, C.byref(v)) == 0
            assert v.value == o.sample()
        elif op == 2:
            e = Ext()
            assert lib.sp1hip_challenger_sample_ext(h, C.byref(e)) == 0
            assert list(e.c) == o.sample_ext().tolist()
        else:
            bits, v = int(rng.integers(1, 31)), C.c_uint32()
            assert lib.sp1hip_challenger_sample_bits(h, bits, C.byref(v)) == 0
            assert v.value == o.sample_bits(bits)
    st = np.zeros(34, np.uint32)
    assert lib.sp1hip_challenger_state(h, st.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    assert np.array_equal(st, o.state())
    # <= 8-bit grinding runs on the host: smallest witness, same post-state as the oracle
    w = C.c_uint32()
    assert lib.sp1hip_challenger_grind(h, 7, C.byref(w), None) == 0
    assert w.value == o.grind(7)
    assert lib.sp1hip_challenger_state(h, st.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    assert np.array_equal(st, o.state())
    lib.sp1hip_challenger_free(h)


def test_library_challenger_replays_reference_transcript(lib):
    """The product's host DuplexChallenger (behind sp1hip_challenger_*) reproduces the reference's real
    shard-proof transcript: grinding witnesses accepted, sumcheck points / betas / query indices equal."""
    import transcript_tape as tt
    from sp1_amd import api
    ch = api.DuplexChallenger()
    n_ops, pinned = tt.replay(ch)
    assert n_ops == len(tt.TAPE["ops"]) and pinned >= 500
    assert np.array_equal(orc.from_monty(ch.state()[:16]), tt.TAPE["final_state"])


def test_product_does_not_reference_the_oracle():
    """The product path may never import, link or call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sp1_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "kb_pcs" not in txt, f
    from sp1_amd import _lib
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True).stdout.decode()
    assert "oracle" not in out


def test_ffi_struct_layouts_match_the_header(tmp_path):
    """The reference keeps an FFI struct-layout test next to its kernels (sp1-gpu/crates/zerocheck/tests/ffi_layout.rs);
    here: every struct of include/sp1hip.h, compiled by the C compiler, has the size and field offsets of its ctypes
    mirror in sp1_amd/_lib.py (the same check a Rust `#[repr(C)]` binding would need)."""
    import re
    import subprocess
    from sp1_amd import _lib
    pairs = {"sp1hip_ext_t": _lib.Ext, "sp1hip_tensor_t": _lib.Tensor, "sp1hip_table_t": _lib.Table,
             "sp1hip_host_table_t": _lib.HostTable, "sp1hip_fri_config_t": _lib.FriConfig, "sp1hip_zc_chip_t": _lib.ZcChip,
             "sp1hip_gkr_chip_t": _lib.GkrChip, "sp1hip_shard_chip_t": _lib.ShardChip, "sp1hip_shard_params_t": _lib.ShardParams,
             "sp1hip_vk_t": _lib.Vk, "sp1hip_pool_chip_t": _lib.PoolChip, "sp1hip_pool_times_t": _lib.PoolTimes,
             "sp1hip_rv64_shard_info_t": _lib.Rv64ShardInfo, "sp1hip_rv64_shard_limits_t": _lib.Rv64ShardLimits,
             "sp1hip_rv64_alu_event_t": _lib.Rv64AluEvent}
    header = open(os.path.join(ROOT, "include", "sp1hip.h")).read()
    declared = set(re.findall(r"}\s*(sp1hip_\w+_t)\s*;", header))
    assert declared == set(pairs), "a struct of the header has no ctypes mirror (or the other way round): %s" % (declared ^ set(pairs))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sp1hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('return 0; }')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    seen = 0
    for line in out:
        if not line:
            continue
        cname, field, value = line.split()
        cls = pairs[cname]
        want = C.sizeof(cls) if field == "size" else getattr(cls, field).offset
        assert int(value) == want, (cname, field, value, want)
        seen += 1
    assert seen == sum(1 + len(c._fields_) for c in pairs.values())


def test_status_constants_match_the_header():
    from sp1_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sp1hip.h")).read()
    codes = dict((k, int(v)) for k, v in re.findall(r"SP1HIP_(\w+)\s*=\s*(-?\d+)", hdr))
    assert codes["SUCCESS"] == _lib.SUCCESS and codes["ERROR_INVALID_ARGUMENT"] == _lib.ERROR_INVALID_ARGUMENT
    assert codes["ERROR_BUFFER_TOO_SMALL"] == _lib.ERROR_BUFFER_TOO_SMALL and codes["ERROR_NOT_READY"] == _lib.ERROR_NOT_READY


def test_vk_observe_into_replays_the_head_of_the_reference_transcript(lib):
    """`sp1hip_vk_observe_into` = MachineVerifyingKey::observe_into (/root/reference/crates/hypercube/src/verifier/config.rs:L97-L112).
    The reference's real (vk, proof) pair starts its transcript with exactly that call: feeding the real vk's fields through
    the entry point must land in the state the tape has when the shard proof starts."""
    import transcript_tape as tt
    from sp1_amd import _lib, api
    T = tt.TAPE
    k = int(T["shard_start_op"])
    ref = orc.Challenger()
    assert tt.replay(ref, stop_before_op=k)[0] == k
    words = []
    for op, arg, off, _ in T["ops"][:k]:
        assert op == 0                                            # the head of the transcript only observes
        words += list(orc.to_monty(T["data"][off:off + arg]))
    assert len(words) == 8 + 3 + 14 + 1 + 6 and not any(words[-6:])
    vk = _lib.Vk()
    vk.preprocessed_commit[:] = words[0:8]
    vk.pc_start[:] = words[8:11]
    vk.initial_global_cumulative_sum[:] = words[11:25]
    vk.enable_untrusted_programs = words[25]
    ch = api.DuplexChallenger()
    assert lib.sp1hip_vk_observe_into(C.byref(vk), ch.h) == 0
    assert np.array_equal(ch.state(), ref.state())


def test_injected_pow_witness_is_used_and_checked(lib):
    """The witness-injection knob: grind returns the SMALLEST witness by default, the caller's when
    one is injected (so a Rust-made proof's `find_any` witness can be replayed), and refuses an invalid one."""
    from sp1_amd import _lib, api
    base = api.DuplexChallenger()
    base.observe(orc.to_monty(np.arange(1, 12, dtype=np.uint32)))
    bits = 6
    valid = []
    w = 0
    while len(valid) < 3:                                         # host path (bits <= 8): no device needed
        if base.clone().check_witness(bits, int(orc.to_monty(np.array([w], np.uint32))[0])):
            valid.append(w)
        w += 1
    def grind(ch):                                                # (api.grind takes a torch stream; no device here)
        out = C.c_uint32()
        st = lib.sp1hip_challenger_grind(ch.h, bits, C.byref(out), None)
        return st, int(orc.from_monty(np.array([out.value], np.uint32))[0])

    a = base.clone()
    assert grind(a) == (0, valid[0])
    b = base.clone()
    b.inject_pow_witnesses([valid[2]])
    assert grind(b) == (0, valid[2])
    assert not np.array_equal(a.state(), b.state())
    assert grind(b)[0] == 0                                       # queue consumed: back to searching
    c = base.clone()
    invalid = next(x for x in range(valid[2]) if x not in valid)
    c.inject_pow_witnesses([invalid])
    assert grind(c)[0] == _lib.ERROR_INVALID_ARGUMENT and b"injected" in lib.sp1hip_last_error()
