"""GPU tests (-m gpu) of the host-trace staging path (staging half): sp1hip_stage_tables turns
row-major host traces into the column-major device tables the prover consumes. Pure data movement, so the
check is exact equality with the numpy transpose — ragged and empty shapes, tables that span several
staging chunks, pinned and pageable sources, two streams at once — and that a shard staged this way
commits to the oracle's commitment."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def col_major_words(cm):
    return cm.words.cpu().numpy().view(np.uint32).reshape(cm.width, cm.height)


SHAPES = [(1000, 8), (1, 1), (0, 5), (64, 64), (65, 63), (12345, 400), (3, 1000), (4097, 1), (63, 129)]


def test_stage_tables_is_the_transpose(api):
    tabs = [orc.random_felts(s, 40 + i) if s[0] else np.zeros(s, np.uint32) for i, s in enumerate(SHAPES)]
    out = api.stage_tables(tabs)
    torch.cuda.synchronize()
    for a, cm in zip(tabs, out):
        assert (cm.height, cm.width) == a.shape
        assert np.array_equal(col_major_words(cm), a.T)


@pytest.mark.parametrize("pinned", [True, False])
def test_stage_tables_multi_chunk(api, pinned):
    # 25.2 M words = 101 MB: four 32 MiB staging chunks, the last one ragged; and a wide table (1 chunk = 64-row tiles)
    shapes = [((1 << 20) + 77, 24), (300, 20000)]
    g = torch.Generator().manual_seed(5)
    hosts = []
    for s in shapes:
        t = torch.randint(0, 0x7F000001, s, dtype=torch.int32, generator=g)
        hosts.append(t.pin_memory() if pinned else t)
    for _ in range(2):                              # second pass re-uses the side stream, events and staging blocks
        out = api.stage_tables(hosts)
        torch.cuda.synchronize()
        for t, cm in zip(hosts, out):
            assert torch.equal(cm.words.view(cm.width, cm.height).cpu(), t.t())


def test_stage_on_two_streams(api):
    shapes = [(200000, 37), (150001, 52)]
    g = torch.Generator().manual_seed(6)
    hosts = [torch.randint(0, 0x7F000001, s, dtype=torch.int32, generator=g).pin_memory() for s in shapes]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for st, h in zip(streams, hosts):
        with torch.cuda.stream(st):
            outs.append(api.stage_tables([h, h], stream=st))
    torch.cuda.synchronize()
    for h, o in zip(hosts, outs):
        for cm in o:
            assert torch.equal(cm.words.view(cm.width, cm.height).cpu(), h.t())


def test_staged_shard_commits_like_the_oracle(api):
    from test_oracle_jagged import make_rounds
    shapes = [[(1 << 10, 3), (777, 5), (0, 2), (33, 7)]]
    L, lsh, batch, lb = 12, 8, 4, 1
    rounds, tabs = make_rounds(shapes, L, lsh, batch, 19, lb)
    staged = api.stage_tables([np.ascontiguousarray(t) for t in tabs[0]])
    commit, sd = api.JaggedProver(L, lsh, batch, lb).commit_multilinears(staged)
    assert np.array_equal(commit, rounds[0].commit)


def test_host_register_then_stage(api):
    """Caller-owned memory pinned in place (sp1hip_host_register, the reference's cuda_host_register) feeds the staging
    path like a pinned allocation does; unregistering afterwards leaves the array usable."""
    a = orc.random_felts((50000, 33), 77)
    L = api._L()
    api.check(L.sp1hip_host_register(a.ctypes.data, a.nbytes))
    try:
        (cm,) = api.stage_tables([a])
        torch.cuda.synchronize()
        assert np.array_equal(col_major_words(cm), a.T)
    finally:
        api.check(L.sp1hip_host_unregister(a.ctypes.data))
    assert int(a[0, 0]) == int(orc.random_felts((50000, 33), 77)[0, 0])
