"""GPU parity (-m gpu) of the zerocheck sumcheck (a9–a12): proof bytes and transcript state equal to the
oracle's on hand-written AIRs, including ragged / odd / empty / full heights and a nonzero padded-row
adjustment; the oracle's restatement of the reference verifier accepts the GPU proof."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import pyoracle as orc  # noqa: E402
from test_oracle_zerocheck import setup  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _gpu_chips(api, chips):
    out = []
    for name, air, main, prep in chips:
        d_main = api.ColMajor.from_row_major_host(main) if main.shape[0] else None
        d_prep = api.ColMajor.from_row_major_host(prep) if prep is not None and prep.shape[0] else None
        out.append(api.ZerocheckChip(air, d_main, d_prep))
    return out


@pytest.mark.parametrize("heights,L", [
    ({"Mul": 8}, 3),
    ({"Mul": 5, "Affine": 3, "Sbox": 6}, 3),
    ({"Affine": 1, "Mul": 1}, 2),
    ({"Affine": 7, "Empty": 0, "Sbox": 2}, 4),
    ({"Mul": 1}, 1),
    ({"Affine": 2}, 1),
    ({"Affine": 1000, "Mul": 4096, "Sbox": 2049, "Sbox2": 1}, 12),
    ({"Affine": 70000, "Mul": 1 << 17, "Sbox": 99999}, 17),          # multi-block sums, grid-stride
    ({"Chain": 300, "Manyregs": 1000, "Mul": 77}, 10),               # chunk-limit overflow, scratch-register tier, TOUCH columns
    ({"Chain": 1, "Manyregs": 2}, 3),
])
def test_zerocheck_matches_oracle(api, heights, L):
    chips, zc, zeta, alpha, gkr, publics, o_ch = setup(heights, L, 40 + L)
    g_ch = api.DuplexChallenger()
    g_ch.observe(orc.random_felts((8,), 40 + L))
    assert np.array_equal(g_ch.sample_point(L), zeta)
    assert np.array_equal(g_ch.sample_ext_element(), alpha) and np.array_equal(g_ch.sample_ext_element(), gkr)
    v_ch = o_ch.clone()
    want = orc.zerocheck_prove(zc, L, zeta, alpha, gkr, publics, o_ch)
    openings = np.concatenate([c.openings for c in zc])
    got = api.zerocheck(_gpu_chips(api, chips), L, zeta, openings, alpha, gkr, publics, g_ch)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())
    assert orc.zerocheck_verify(zc, [c.real_rows for c in zc], L, zeta, alpha, gkr, publics, got, v_ch) == 0


def test_zerocheck_rejects_bad_programs_and_keeps_transcript(api):
    chips, zc, zeta, alpha, gkr, publics, _ = setup({"Mul": 4}, 2, 3)
    g = _gpu_chips(api, chips)
    g[0].program = g[0].program.copy()
    g[0].program[0, 1] = 99                       # LOAD_MAIN column out of range
    ch = api.DuplexChallenger()
    before = ch.state()
    with pytest.raises(api._lib.Sp1HipError):
        api.zerocheck(g, 2, zeta, np.concatenate([c.openings for c in zc]), alpha, gkr, publics, ch)
    assert np.array_equal(ch.state(), before)
