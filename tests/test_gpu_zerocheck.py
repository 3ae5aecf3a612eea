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


@pytest.mark.parametrize("heights,L", [
    ({"Affine": 1000, "Mul": 4096, "Sbox": 2049, "Sbox2": 1}, 12),
    ({"Chain": 300, "Manyregs": 1000, "Mul": 77}, 10),
])
def test_zerocheck_programs_staged_in_lds_give_the_same_bytes(api, monkeypatch, heights, L):
    """Constraint programs stream through the scalar cache by default; SP1HIP_ZC_STAGE_MAX stages the short ones in LDS
    (the other instruction-fetch path of the interpreter)."""
    monkeypatch.setenv("SP1HIP_ZC_STAGE_MAX", "1024")
    chips, zc, zeta, alpha, gkr, publics, o_ch = setup(heights, L, 40 + L)
    g_ch = api.DuplexChallenger()
    g_ch.observe(orc.random_felts((8,), 40 + L))
    g_ch.sample_point(L); g_ch.sample_ext_element(); g_ch.sample_ext_element()
    want = orc.zerocheck_prove(zc, L, zeta, alpha, gkr, publics, o_ch)
    got = api.zerocheck(_gpu_chips(api, chips), L, zeta, np.concatenate([c.openings for c in zc]), alpha, gkr, publics, g_ch)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())


@pytest.mark.parametrize("env", [{"SP1HIP_ZC_BIVARIATE": "0"}, {"SP1HIP_ZC_FORK": "0"}, {"SP1HIP_ZC_BIVARIATE": "0", "SP1HIP_ZC_FORK": "0"}])
@pytest.mark.parametrize("heights,L", [
    ({"Mul": 5, "Affine": 3, "Sbox": 6}, 3),
    ({"Affine": 1000, "Mul": 4096, "Sbox": 2049, "Sbox2": 1}, 12),
    ({"Chain": 300, "Manyregs": 1000, "Mul": 77}, 10),
])
def test_zerocheck_sequential_rounds_and_single_stream_give_the_same_bytes(api, monkeypatch, env, heights, L):
    """The default path proves rounds 0 and 1 from one pass over the base-field traces (bivariate grid) and spreads a round's
    launches over fork streams; SP1HIP_ZC_BIVARIATE=0 / SP1HIP_ZC_FORK=0 select the sequential rounds / one stream: same bytes."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    chips, zc, zeta, alpha, gkr, publics, o_ch = setup(heights, L, 40 + L)
    g_ch = api.DuplexChallenger()
    g_ch.observe(orc.random_felts((8,), 40 + L))
    g_ch.sample_point(L); g_ch.sample_ext_element(); g_ch.sample_ext_element()
    want = orc.zerocheck_prove(zc, L, zeta, alpha, gkr, publics, o_ch)
    got = api.zerocheck(_gpu_chips(api, chips), L, zeta, np.concatenate([c.openings for c in zc]), alpha, gkr, publics, g_ch)
    assert got == want
    assert np.array_equal(g_ch.state(), o_ch.state())


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


@pytest.mark.parametrize("rows,width,pad", [(1000, 5, True), (1001, 3, True), (1, 2, True), (777, 4, False), (2, 1, False)])
def test_fix_last_variable_export(api, rows, width, pad):
    """sp1hip_fix_last_variable (restrict.rs mle_fix_last_variable): out[i][c] = x + alpha (y - x) with the column's
    padding value for a missing last row; base input, then the extension output fed back in (extension padding).
    Checked against the pure-Python extension arithmetic of oracle/kb_py.py in the canonical domain."""
    import ctypes as C
    import kb_py
    from sp1_amd._lib import Ext
    L = api._L()
    tab = orc.random_felts((rows, width), 31 + rows)               # row-major host, Montgomery
    alpha_m = orc.random_felts((4,), 5)
    padding_m = orc.random_felts((width,), 9) if pad else None
    d_in = api.ColMajor.from_row_major_host(tab)
    out_rows = (rows + 1) // 2
    d_out = api.device_words(out_rows * width * 4)
    d_pad = api.to_device(padding_m) if pad else None
    a = Ext()
    for k in range(4):
        a.c[k] = int(alpha_m[k])
    api.check(L.sp1hip_fix_last_variable(api._dptr(d_in.words), rows, width, 0, a, api._dptr(d_pad) if pad else None,
                                         api._dptr(d_out), api._stream_ptr()))
    got = orc.from_monty(api.to_host(d_out)).reshape(width, 4, out_rows)
    t, al = orc.from_monty(tab).astype(object), [int(v) for v in orc.from_monty(alpha_m)]
    pv = [int(v) for v in orc.from_monty(padding_m)] if pad else [0] * width
    want = np.zeros((width, 4, out_rows), dtype=object)
    for c in range(width):
        for i in range(out_rows):
            x = int(t[2 * i, c])
            y = int(t[2 * i + 1, c]) if 2 * i + 1 < rows else pv[c]
            want[c, :, i] = kb_py.ext_add(kb_py.ext_scale(al, (y - x) % kb_py.P), kb_py.ext_from_base(x))
    assert np.array_equal(got.astype(object), want)
    # second application on the extension table, extension padding values
    rows2, out2 = out_rows, (out_rows + 1) // 2
    beta_m = orc.random_felts((4,), 6)
    pad2_m = orc.random_felts((width * 4,), 10) if pad else None
    d_pad2 = api.to_device(pad2_m) if pad else None
    d_out2 = api.device_words(out2 * width * 4)
    b = Ext()
    for k in range(4):
        b.c[k] = int(beta_m[k])
    api.check(L.sp1hip_fix_last_variable(api._dptr(d_out), rows2, width, 1, b, api._dptr(d_pad2) if pad else None,
                                         api._dptr(d_out2), api._stream_ptr()))
    got2 = orc.from_monty(api.to_host(d_out2)).reshape(width, 4, out2)
    be = [int(v) for v in orc.from_monty(beta_m)]
    p2 = orc.from_monty(pad2_m).reshape(width, 4) if pad else np.zeros((width, 4), np.uint32)
    want2 = np.zeros((width, 4, out2), dtype=object)
    for c in range(width):
        for i in range(out2):
            x = [int(v) for v in want[c, :, 2 * i]]
            y = [int(v) for v in want[c, :, 2 * i + 1]] if 2 * i + 1 < rows2 else [int(v) for v in p2[c]]
            want2[c, :, i] = kb_py.ext_add(kb_py.ext_mul(be, kb_py.ext_sub(y, x)), x)
    assert np.array_equal(got2.astype(object), want2)
