"""GPU parity (-m gpu) of the BabyBear instantiation of the commit path (BASELINE config 2 names both fields): RS encode +
Poseidon2 Merkle commitment over p = 2^31 - 2^27 + 1 against the BabyBear oracle (oracle/bb_commit.hpp). The oracle is
pinned by the reference tree for the field / NTT conventions / round constants / sponge, and UNPINNED for the internal
diffusion matrix (it lives in the un-vendored p3-baby-bear; the published convention of that version is used, see the
oracle's header and DESIGN.md §2)."""
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


def test_babybear_permutation_matches_the_oracle(api):
    states = orc.bb_random_felts((4096, 16), 7)
    states[0] = 0
    assert np.array_equal(api.bb_poseidon2_permute(states), orc.bb_permute(states))


@pytest.mark.parametrize("lg_n,widths,lb", [(0, [1], 0), (3, [5, 3], 1), (10, [25] * 10, 2), (13, [32, 7, 1], 2), (17, [9], 1)])
def test_babybear_commit_small_shapes(api, lg_n, widths, lb):
    """incl. the reference's own test shape (10 tensors of 2^10 x 25) and ragged widths (sponge tails)."""
    ms = [orc.bb_random_felts((1 << lg_n, w), 100 + k) for k, w in enumerate(widths)]
    want_c, want_cw, want_tree = orc.bb_commit_mles(ms, lb, True, True)
    commit, cws, tree = api.bb_commit_mles([api.ColMajor.from_row_major_host(m) for m in ms], lb)
    assert np.array_equal(commit, want_c)
    for k in range(len(ms)):
        assert np.array_equal(cws[k].to_row_major_host(), want_cw[k]), k
    assert np.array_equal(api.to_host(tree, want_tree.shape), want_tree)


@pytest.mark.parametrize("W,lb", [(32, 2), (32, 1), (256, 2), (256, 1)])
def test_baseline_config2_babybear_commit_is_bit_exact_against_the_oracle(api, W, lb):
    """SURVEY 8(d) config 2 over BabyBear: n = 2^20 rows, W in {32, 256} as 32-column tensors, log_blowup in {1, 2}, the
    documented generator. Commitment always; full codewords and every Merkle layer for W = 32."""
    ms = [orc.bb_random_felts((1 << 20, 32), 4300 + 17 * k + lb) for k in range(W // 32)]
    want_c, want_cw, want_tree = orc.bb_commit_mles(ms, lb, W == 32, W == 32)
    commit, cws, tree = api.bb_commit_mles([api.ColMajor.from_row_major_host(m) for m in ms], lb)
    assert np.array_equal(commit, want_c)
    if W == 32:
        assert np.array_equal(cws[0].to_row_major_host(), want_cw[0])
        assert np.array_equal(api.to_host(tree, want_tree.shape), want_tree)
