"""CPU checks of the compiled zerocheck kernels (sp1_amd/csrc/zc_jit.hip): the generated source is deterministic, compiles
for gfx950 (hipcc cross-compiles without a GPU), the headers embedded in the library are current, and
__graft_entry__.prebuild_zc_kernels() leaves one code object per eligible chip of the machines this repository proves. The
GPU side — compiled and interpreted rounds give the same proof bytes — is tests/test_gpu_zc_jit.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import zc_airs  # noqa: E402


def test_embedded_headers_are_current():
    import __graft_entry__ as g
    path = os.path.join(g.CSRC, "zc_jit_headers.inc")
    before = open(path).read()
    g.write_jit_headers()
    assert open(path).read() == before, "sp1_amd/csrc/zc_jit_headers.inc is stale: run __graft_entry__.build_hip()"
    for name in ("kb31.hpp", "zc_device.hpp"):
        assert open(os.path.join(g.CSRC, name)).read() in before


def test_generated_source_compiles_for_gfx950(tmp_path):
    import __graft_entry__ as g
    g.build_hip()
    from sp1_amd import api
    from sp1_amd.air import AirProgram
    air = AirProgram("T", 6, 2, cse=True)
    a, b, c, d = (air.main(k) for k in range(4))
    air.assert_zero(a * b - c)
    air.assert_zero(d * (d - 1) * a + air.prep(0) * 7 - air.public(1))
    air.assert_zero((a + 3) * (b - 5) - air.main(4) * 2)
    src, h = api.zerocheck_codegen(air)
    src2, h2 = api.zerocheck_codegen(air)
    assert (src, h) == (src2, h2) and h != 0
    assert 'extern "C" __global__' in src and "zc_jit_first" in src and "zc_jit_ext" in src
    assert src.count("C = kb::ext_add(C,") == 3                       # one accumulation per constraint
    assert "d.main, 5u" in src                                        # the column no constraint reads still enters the batching term
    # a different program hashes differently
    air.assert_zero(a - b)
    assert api.zerocheck_codegen(air)[1] != h
    p = tmp_path / "k.hip"
    p.write_text(src)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--genco", "-I" + g.CSRC,
                           str(p), "-o", str(tmp_path / "k.hsaco")])
    assert (tmp_path / "k.hsaco").stat().st_size > 1000


def test_prebuilt_cache_covers_the_known_machines():
    import __graft_entry__ as g
    g.build_hip()
    n_wanted, _ = g.prebuild_zc_kernels()
    cache = os.path.join(ROOT, "sp1_amd", "lib", "zc_cache")
    objs = [f for f in os.listdir(cache) if f.endswith(".hsaco")]
    assert n_wanted >= 25 and len(objs) == n_wanted
    # the recursion machine's narrow chips are in, its wide Poseidon2 chip is not (it stays interpreted)
    from sp1_amd import api
    from sp1_amd.machines import recursion as R
    have = {f.split(".")[0] for f in objs}
    names = {a.name: ("%016x" % api.zerocheck_codegen(a)[1]) in have for a, _ in R.compress_machine() if a.num_constraints}
    assert names["BaseAlu"] and names["ExtAlu"] and names["Select"] and not names["Poseidon2WideDeg3"]
