"""GPU (-m gpu): the COMPILED zerocheck kernels (zc_jit.hip) against the bytecode interpreter. A shard whose chips are
tall enough for the compiled path is proven with SP1HIP_ZC_JIT=0 (interpreter only), then with the compiled kernels —
compiled by the library's background hipcc into an empty cache directory, waited for — and with the kernels __graft_entry__
prebuilt: identical bytes every time, and the launch counter shows the compiled path really ran."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bench"))

import pyoracle as orc  # noqa: E402

CORE_AREA = (1 << 28) + (1 << 27)


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _prove(api, chips, prep, commit, L, lsh, fri):
    ch = api.DuplexChallenger()
    ch.observe(commit)
    return api.prove_shard(chips, [], prep, L, lsh, 32, ch, *fri), ch.state()


def test_compiled_rounds_give_the_interpreters_bytes(api, monkeypatch, tmp_path):
    from core_shard import build_core_shard
    k = 2                                              # 1/16 of CORE: most chips have >= 2048 rows
    L, lsh, fri = 22 - k, 21 - k, (1, 6, 4)
    chips, meta = build_core_shard(CORE_AREA >> (2 * k), L)
    jp = api.JaggedProver(L, lsh, 32, fri[0])
    commit, prep = jp.commit_multilinears([c[3] for c in chips if c[3] is not None])
    monkeypatch.setenv("SP1HIP_ZC_JIT", "0")
    want, want_state = _prove(api, chips, prep, commit, L, lsh, fri)
    before = api.zerocheck_jit_stats()["launches"]
    assert _prove(api, chips, prep, commit, L, lsh, fri)[0] == want and api.zerocheck_jit_stats()["launches"] == before
    # the kernels prebuilt next to the library
    monkeypatch.setenv("SP1HIP_ZC_JIT", "1")
    got, state = _prove(api, chips, prep, commit, L, lsh, fri)
    st = api.zerocheck_jit_stats()
    assert got == want and np.array_equal(state, want_state)
    assert st["launches"] > before and st["failed"] == 0, st
    v = orc.Challenger()
    v.observe(commit)
    shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
              for a, i, _, _ in chips]
    assert orc.shard_verify(shapes, commit, got, L, lsh, v, *fri) == 0


def test_background_compiler_fills_an_empty_cache(api, monkeypatch, tmp_path):
    """A program the prebuilt cache does not know: the first proof runs interpreted and queues the compile (hipcc on the
    box, in the background), sp1hip_zerocheck_jit_wait blocks until the code object exists, the next proof launches it."""
    from sp1_amd.air import AirProgram, InteractionProgram, VCol
    monkeypatch.setenv("SP1HIP_ZC_JIT", "1")
    monkeypatch.setenv("SP1HIP_CACHE_DIR", str(tmp_path))
    rows, L, lsh, fri = 6000, 13, 12, (1, 5, 4)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(int.from_bytes(os.urandom(4), "little"))
    salt = int(torch.randint(1, api.P, (1,), generator=gen, device="cuda").item())      # a fresh program every run
    air = AirProgram("Fresh", 4, 1, cse=True)
    a, b, c, d = (air.main(k) for k in range(4))
    air.assert_zero(a * b - c)
    air.assert_zero((d * (d - 1)) * (a + salt))
    inter = InteractionProgram("Fresh", 4, 1)
    inter.send(5, [VCol.main(0), VCol.main(1)], VCol.main(3))
    inter.receive(5, [VCol.main(0), VCol.main(1)], VCol.main(3))
    from synthetic_shard import wide_trace
    main = wide_trace(rows, 4, gen)
    prep_t = api.ColMajor(torch.randint(0, api.P, (rows,), dtype=torch.int32, device="cuda", generator=gen), rows, 1)
    chips = [(air, inter, main, prep_t)]
    jp = api.JaggedProver(L, lsh, 32, fri[0])
    commit, prep = jp.commit_multilinears([prep_t])
    s0 = api.zerocheck_jit_stats()
    first, _ = _prove(api, chips, prep, commit, L, lsh, fri)
    assert api.zerocheck_jit_wait(120000) == 0
    s1 = api.zerocheck_jit_stats()
    assert s1["ready"] == s0["ready"] + 1 and s1["failed"] == s0["failed"], (s0, s1)
    assert any(f.endswith(".hsaco") for f in os.listdir(os.path.join(str(tmp_path), "zc")))
    second, _ = _prove(api, chips, prep, commit, L, lsh, fri)
    assert second == first and api.zerocheck_jit_stats()["launches"] > s1["launches"]
    monkeypatch.setenv("SP1HIP_ZC_JIT", "0")
    assert _prove(api, chips, prep, commit, L, lsh, fri)[0] == first
