"""GPU (-m gpu): the prover pool of the C ABI (sp1hip_pool_*): N shards in flight on one GPU, staging included.
Proofs from the pool — traces resident in HBM or staged from pinned host memory, 1 to 3 slots, shards interleaved — are
byte-identical to `sp1hip_prove_shard_with_pk` called directly; a failing shard reports its error through its ticket and
the pool keeps serving."""
import copy
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bench"))


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


@pytest.fixture(scope="module")
def shards(api):
    """Two different shards of the same machine (same chips, same preprocessed table, different main traces)."""
    from synthetic_shard import build_shard
    L, lsh = 13, 12
    area = ((1 << 28) + (1 << 27)) >> 18
    chips_a, prep, _, _ = build_shard(L, lsh, area, seed=5)
    chips_b, prep_b, _, _ = build_shard(L, lsh, area, seed=6)
    # one proving key: shard b uses a's preprocessed table
    chips_b = [(a, i, m, (prep if p is not None else None)) for (a, i, m, p) in chips_b]
    pk = api.ProvingKey([prep], L, lsh, 32, log_blowup=1, num_queries=7, pow_bits=5)
    want = [pk.prove_shard(chips_a, []), pk.prove_shard(chips_b, [])]
    assert want[0] != want[1]
    torch.cuda.synchronize()
    return pk, [chips_a, chips_b], want


@pytest.mark.parametrize("n_slots", [1, 2, 3])
def test_pool_proofs_equal_direct_proofs(api, shards, n_slots):
    pk, chips, want = shards
    pool = api.ProverPool(n_slots)
    order = [0, 1, 1, 0, 1, 0, 0]
    tickets = [pool.submit(pk, chips[k]) for k in order]
    slots = set()
    for t, k in zip(tickets, order):
        proof, times = pool.wait(t)
        assert proof == want[k]
        assert times["proving_ms"] > 0 and 0 <= times["slot"] < n_slots
        slots.add(times["slot"])
    assert len(slots) == min(n_slots, len(slots)) and (n_slots == 1 or len(slots) >= 2)
    pool.close()


def test_pool_stages_host_traces(api, shards):
    """Main traces handed over as pinned ROW-major host words: the pool uploads + transposes them (sp1hip_stage_tables on
    its stager stream) while other shards are being proven; same bytes."""
    pk, chips, want = shards
    host = []
    for k in range(2):
        host.append([(a, i, api.PinnedHost(m.to_row_major_host()) if m is not None else None, p) for (a, i, m, p) in chips[k]])
    pool = api.ProverPool(2)
    order = [1, 0, 0, 1, 1]
    tickets = [pool.submit(pk, host[k]) if j % 2 == 0 else pool.submit(pk, chips[k]) for j, k in enumerate(order)]
    for t, k in zip(tickets, order):
        proof, times = pool.wait(t)
        assert proof == want[k]
    pool.close()


def test_pool_reports_a_failed_shard_and_keeps_serving(api, shards):
    pk, chips, want = shards
    bad = list(chips[0])
    air = copy.deepcopy(bad[0][0])
    air.instrs[0] = (0, 9999, 0)                     # LOAD_MAIN of a column that does not exist
    air._array_cache = None
    bad[0] = (air, bad[0][1], bad[0][2], bad[0][3])
    pool = api.ProverPool(2)
    t_bad, t_good = pool.submit(pk, bad), pool.submit(pk, chips[1])
    with pytest.raises(api._lib.Sp1HipError) as e:
        pool.wait(t_bad)
    assert "column" in str(e.value) or "program" in str(e.value)
    assert pool.wait(t_good)[0] == want[1]
    with pytest.raises(api._lib.Sp1HipError):        # a ticket is collected once
        pool.wait(t_good)
    # try_wait: None while in flight, the proof afterwards
    t = pool.submit(pk, chips[0])
    got = None
    for _ in range(200000):
        got = pool.wait(t, block=False)
        if got is not None:
            break
    assert got is not None and got[0] == want[0]
    pool.close()
