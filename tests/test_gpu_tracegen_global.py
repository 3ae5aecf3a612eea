"""GPU (-m gpu): device trace generation for the RISC-V Global chip (sp1hip_tracegen_riscv_global) — from the
(message, is_receive, kind) events alone the device table equals the host trace of riscv_trace.py / septic.py word for word
(Poseidon2 columns, lifted curve points, offsets, range-check bytes, the running digest sum, the padding rows), and a shard proven
from the device-generated table equals the proof from the host table."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from sp1_amd.machines import riscv as R  # noqa: E402
from sp1_amd.machines import riscv_trace as RT  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def _events(main, n):
    """[n, 9] int32 events from the host trace's message / is_receive / kind columns."""
    L = R.chip("Global")[0].layout
    ev = torch.zeros((n, 9), dtype=torch.int64, device=main.device)
    ev[:, :8] = main[:n, L["message"]:L["message"] + 8]
    ev[:, 8] = main[:n, L["is_receive"]] | (main[:n, L["kind"]] << 8)
    return ev.to(torch.int32).contiguous()


@pytest.mark.parametrize("counts,K", [({"Add": 4, "LoadWord": 6, "StoreWord": 6, "UType": 8, "Ecall": 6}, 2),
                                      ({"Add": 40, "Addi": 40, "LoadByte": 60, "LoadDouble": 60, "StoreByte": 60, "StoreDouble": 60, "UType": 30, "Branch": 20}, 3)])
def test_device_global_trace_equals_the_host_trace(api, counts, K):
    import core_real
    machine, tabs, _ = RT.generate(counts, K=K, seed=17, device="cuda")
    main = tabs["Global"][1]
    n = int((main[:, R.chip("Global")[0].layout["is_real"]] == 1).sum())
    assert 0 < n <= main.shape[0] and main.shape[0] > n                  # padding rows are part of the comparison
    got = api.tracegen_riscv_global(_events(main, n), main.shape[0])
    want = core_real.to_col_major(main)
    assert got.height == want.height and got.width == want.width
    g, w = got.words.view(got.width, got.height), want.words.view(want.width, want.height)
    bad = (g != w).nonzero()
    assert bad.numel() == 0, ("first mismatch (column, row):", bad[0].tolist())


def test_shard_proof_from_the_device_generated_global_table(api):
    import core_real
    counts = {"Add": 4, "Addi": 6, "LoadWord": 6, "StoreWord": 6, "UType": 8, "Branch": 4}
    machine, tabs, publics = RT.generate(counts, K=2, seed=19, device="cuda")
    dev = [(a, i, core_real.to_col_major(tabs[a.name][1]), core_real.to_col_major(tabs[a.name][0]) if tabs[a.name][0] is not None else None)
           for a, i in machine]
    L, lsh, batch = 17, 12, 8
    commit, prep = api.JaggedProver(L, lsh, batch, 1).commit_multilinears([d[3] for d in dev if d[3] is not None])

    def prove(chips):
        ch = api.DuplexChallenger()
        ch.observe(commit)
        return api.prove_shard(chips, RT.to_monty_np(publics), prep, L, lsh, batch, ch, 1, 5, 4)
    want = prove(dev)
    main = tabs["Global"][1]
    n = int((main[:, R.chip("Global")[0].layout["is_real"]] == 1).sum())
    table = api.tracegen_riscv_global(_events(main, n), main.shape[0])
    dev2 = [(a, i, table if a.name == "Global" else m, p) for a, i, m, p in dev]
    assert prove(dev2) == want
