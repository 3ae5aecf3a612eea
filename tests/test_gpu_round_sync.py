"""Stress of the sumcheck-round tail (`rs_finish`, sp1_amd/csrc/round_sync.hpp): the fence-free hand-over between
workgroups (coherent sc1 accesses ordered by s_waitcnt vmcnt(0), two-level ticket) — thousands of launches at grid sizes
around the ticket's group boundaries, every published sum checked on the host; then the other cross-workgroup hand-over
to the host, a LogUp-GKR layer's last fold storing its rows straight into mapped memory (tests/native/rs_finish_stress.hip)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "native", "rs_finish_stress")


@pytest.mark.gpu
def test_rs_finish_stress():
    assert os.path.exists(EXE), "tests/native/rs_finish_stress is not built (__graft_entry__.build())"
    out = subprocess.run([EXE, "4000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 wrong, 0 counter words left non-zero" in out.stdout, out.stdout
    assert "direct rows: 4000 launches, 0 wrong words" in out.stdout, out.stdout
