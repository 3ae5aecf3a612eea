"""GPU parity (-m gpu) of LogUp-GKR: bincode(LogupGkrProof) and the transcript state equal
the oracle's (which materialises every layer densely, an independent algorithm) byte for byte; the oracle's
restated reference verifier, including the final interaction check, accepts the GPU proof."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from gkr_chips import make_gkr_chips  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _dev(api, chips):
    out = []
    for prog, main, prep in chips:
        d_main = api.ColMajor.from_row_major_host(main) if main.shape[0] else None
        d_prep = api.ColMajor.from_row_major_host(prep) if prep is not None and prep.shape[0] else None
        out.append((prog, d_main, d_prep))
    return out


@pytest.mark.parametrize("n_tuples,L,with_empty,dup", [
    (4, 3, False, 2),
    (5, 4, True, 3),
    (1, 1, False, 2),
    (2, 2, False, 1),          # one layer of one row variable: passes (sum 1), (fold 1)
    (3, 5, False, 1),
    (9, 6, True, 2),           # odd and even numbers of row variables per layer: (2,2) / (2,1) / (2,0) / (1,0) passes
    (37, 7, True, 3),          # odd live counts at several levels: partial grid cells
    (300, 10, True, 3),        # multi-tile rows
    (1300, 12, True, 3),       # several complete lane-blocked blocks per table
])
def test_gkr_proof_matches_oracle(api, n_tuples, L, with_empty, dup):
    chips = make_gkr_chips(n_tuples, 10 + L, with_empty, dup)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    seed = orc.random_felts((9,), L)
    o_ch.observe(seed)
    g_ch.observe(seed)
    v_ch = o_ch.clone()
    want = orc.gkr_prove(chips, L, o_ch)
    got = api.logup_gkr(_dev(api, chips), L, g_ch)
    assert len(got) == len(want)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    assert orc.gkr_verify(chips, [c[1].shape[0] for c in chips], L, got, v_ch) == 0
    point, opened = api.parse_logup_gkr_proof(got)
    assert point.shape == (L, 4) and [o[0] for o in opened] == [c[0].name for c in chips]


@pytest.mark.parametrize("flat_slots", ["0", "64", "1024"])     # 0: every pass one workgroup per tile; 64 / 1024: both forms inside one layer
@pytest.mark.parametrize("n_tuples,L,with_empty,dup", [(37, 7, True, 3), (300, 10, True, 3), (1300, 12, True, 3)])
def test_gkr_small_pass_forms_give_the_same_bytes(api, monkeypatch, n_tuples, L, with_empty, dup, flat_slots):
    """Passes with few rows run one row per lane of a flat launch (SP1HIP_GKR_FLAT_SLOTS, default 65536 — every pass of
    these sizes); forcing the tiled form, or a mix, must not change a byte."""
    monkeypatch.setenv("SP1HIP_GKR_FLAT_SLOTS", flat_slots)
    chips = make_gkr_chips(n_tuples, 10 + L, with_empty, dup)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    seed = orc.random_felts((9,), L)
    o_ch.observe(seed)
    g_ch.observe(seed)
    want = orc.gkr_prove(chips, L, o_ch)
    got = api.logup_gkr(_dev(api, chips), L, g_ch)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())


@pytest.mark.parametrize("n_tuples,L,with_empty,dup", [(5, 4, True, 3), (37, 7, True, 3), (300, 10, True, 3), (1300, 12, True, 3)])
def test_gkr_scalar_host_rounds_give_the_same_bytes(api, monkeypatch, n_tuples, L, with_empty, dup):
    """The interaction-variable rounds run in AVX-512 on the host where the CPU has it (gkr_host.cpp; the other GPU tests);
    SP1HIP_HOST_SIMD=0 forces the scalar rounds on the helper threads — the path of CPUs without AVX-512."""
    monkeypatch.setenv("SP1HIP_HOST_SIMD", "0")
    chips = make_gkr_chips(n_tuples, 10 + L, with_empty, dup)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    seed = orc.random_felts((9,), L)
    o_ch.observe(seed)
    g_ch.observe(seed)
    want = orc.gkr_prove(chips, L, o_ch)
    got = api.logup_gkr(_dev(api, chips), L, g_ch)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())


@pytest.mark.parametrize("n_tuples,L,with_empty,dup", [(4, 3, False, 2), (3, 5, False, 1), (9, 6, True, 2), (37, 7, True, 3), (300, 10, True, 3), (1300, 12, True, 3)])
def test_gkr_separate_tree_kernels_give_the_same_bytes(api, monkeypatch, n_tuples, L, with_empty, dup):
    """The first layer and the two levels below it come out of one pass over the traces, the rest of the fraction tree two levels
    per launch (first_layers_kernel / transition2_kernel, round 6; every other test runs them: heights 8 .. 3900, multiples of four
    and not, partial quads at every level). SP1HIP_GKR_FUSED=0 builds the same tree level by level (first_layer_kernel +
    transition_kernel): not a byte may differ."""
    monkeypatch.setenv("SP1HIP_GKR_FUSED", "0")
    chips = make_gkr_chips(n_tuples, 10 + L, with_empty, dup)
    o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
    seed = orc.random_felts((9,), L)
    o_ch.observe(seed)
    g_ch.observe(seed)
    want = orc.gkr_prove(chips, L, o_ch)
    got = api.logup_gkr(_dev(api, chips), L, g_ch)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())


@pytest.mark.parametrize("n_tuples,L,with_empty,dup", [(9, 6, True, 2), (300, 10, True, 3), (1300, 12, True, 3)])
def test_gkr_passes_behind_the_host_gate_give_the_same_bytes(api, monkeypatch, n_tuples, L, with_empty, dup):
    """SP1HIP_GATE=1: the small passes are enqueued one hand-over early and take their challenges from mapped host memory once the
    host has written them (HostGate, round_sync.hpp: one polling lane per workgroup, bounded by the wall clock) instead of from
    their kernel arguments — an A/B knob that measured no faster (gkr.hip). The bytes and the transcript must not change, and
    the gate must leave no launch waiting (the second proof on the same stream runs)."""
    monkeypatch.setenv("SP1HIP_GATE", "1")
    chips = make_gkr_chips(n_tuples, 10 + L, with_empty, dup)
    for _ in range(2):
        o_ch, g_ch = orc.Challenger(), api.DuplexChallenger()
        seed = orc.random_felts((9,), L)
        o_ch.observe(seed)
        g_ch.observe(seed)
        want = orc.gkr_prove(chips, L, o_ch)
        got = api.logup_gkr(_dev(api, chips), L, g_ch)
        assert got == want
        assert np.array_equal(g_ch.state(), o_ch.state())


def test_gkr_rejects_unsorted_chips_and_keeps_transcript(api):
    chips = make_gkr_chips(4, 3)
    dev = _dev(api, chips)
    ch = api.DuplexChallenger()
    before = ch.state()
    with pytest.raises(api._lib.Sp1HipError):
        api.logup_gkr(dev[::-1], 3, ch)
    assert np.array_equal(ch.state(), before)
