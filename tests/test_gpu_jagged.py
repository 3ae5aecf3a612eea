"""GPU parity (-m gpu) of the jagged PCS evaluation proof: bincode(JaggedPcsProof) and
the transcript state equal the oracle's byte for byte; the oracle's restatement of the reference
verifier (the one that accepts the reference's real proof) accepts the GPU proof."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from test_oracle_jagged import CASES, claims_for, make_rounds  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


BIG = [
    ([[(1 << 10, 3), (777, 5), (0, 2), (33, 7)], [(1 << 12, 9), (4095, 2), (1, 1), (2048, 30)]], 12, 8, 4),
    ([[(70000, 3)], [(1 << 17, 5), (99999, 4), (123, 40)]], 17, 12, 8),       # multi-block sums, odd areas
    # every column starts at a multiple of 64 / 32 / 4 in the dense order: the first 6 / 5 / 2 folds keep J factored
    # (jg_foldf_sum), then the prover leaves the factored form (jg_materialize_j)
    ([[(1 << 10, 3), (768, 5), (0, 2), (64, 7)], [(1 << 12, 9), (4032, 2), (128, 1), (2048, 30)]], 12, 8, 4),
    ([[(32, 3), (96, 1)], [(1 << 15, 5), (65504, 4), (160, 40)]], 16, 10, 8),
    ([[(1 << 10, 3), (772, 5), (12, 7)]], 10, 6, 4),
]


@pytest.mark.parametrize("factored", ["1", "0"])
@pytest.mark.parametrize("shapes,L,lsh,batch", CASES + BIG)
def test_jagged_proof_matches_oracle(api, monkeypatch, shapes, L, lsh, batch, factored):
    monkeypatch.setenv("SP1HIP_JAGGED_FACTORED", factored)       # "0": always materialise the j tables
    lb, nq, pw = 1, 6, 4
    rounds, tabs = make_rounds(shapes, L, lsh, batch, 7 + L, lb)
    jp = api.JaggedProver(L, lsh, batch, lb)
    g_rounds, g_commits = [], []
    for tb in tabs:
        dev = [api.ColMajor.from_row_major_host(t) if t.shape[0] else api.ColMajor(torch.zeros(0, dtype=torch.int32, device="cuda"), 0, t.shape[1])
               for t in tb]
        c, sd = jp.commit_multilinears(dev)
        g_rounds.append(sd)
        g_commits.append(c)
    for r, c in zip(rounds, g_commits):
        assert np.array_equal(r.commit, c)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    for c in g_commits:
        o_ch.observe(c)
        g_ch.observe(c)
    z_row = o_ch.sample_point(L)
    assert np.array_equal(g_ch.sample_point(L), z_row)
    claims = claims_for(tabs, L, z_row)
    v_ch = o_ch.clone()
    want = orc.jagged_prove(z_row, claims, rounds, lsh, o_ch, lb, nq, pw)
    got = jp.prove_trusted_evaluations(z_row, claims, g_rounds, g_ch, nq, pw)
    assert len(got) == len(want)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    assert orc.jagged_verify(g_commits, z_row, claims, got, lsh, v_ch, lb, nq, pw) == 0


@pytest.mark.parametrize("shapes,L,lsh,batch", [BIG[2], BIG[3]])
def test_jagged_two_pass_form_of_rounds_0_and_1_gives_the_same_bytes(api, monkeypatch, shapes, L, lsh, batch):
    """With every table height a multiple of 8 the prover takes rounds 0 and 1 of the jagged sumcheck from ONE pass over the
    base words (jg_round01_tables; the default, which test_jagged_proof_matches_oracle covers on these shapes);
    SP1HIP_JAGGED_LOOKAHEAD=0 keeps the two-pass form."""
    monkeypatch.setenv("SP1HIP_JAGGED_LOOKAHEAD", "0")
    test_jagged_proof_matches_oracle(api, monkeypatch, shapes, L, lsh, batch, "1")


def test_jagged_prove_rejects_bad_input_and_keeps_transcript(api):
    shapes, L, lsh, batch = CASES[0]
    _, tabs = make_rounds(shapes, L, lsh, batch, 3)
    jp = api.JaggedProver(L, lsh, batch, 1)
    dev = [api.ColMajor.from_row_major_host(t) for t in tabs[0]]
    _, sd = jp.commit_multilinears(dev)
    ch = api.DuplexChallenger()
    before = ch.state()
    z = orc.random_felts((L, 4), 1)
    with pytest.raises(api._lib.Sp1HipError):
        jp.prove_trusted_evaluations(z, [np.zeros((1, 4), np.uint32)], [sd], ch, 6, 4)      # wrong number of claims
    assert np.array_equal(ch.state(), before)
