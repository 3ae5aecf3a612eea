"""Host helper threads (sp1_amd/csrc/host_par.hpp): the fork/join used by the host arithmetic between device hand-overs."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("threads", ["1", "3", "8"])
def test_host_par_jobs_match_serial_sums(threads):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "stress")
        subprocess.check_call([gxx, "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "sp1_amd", "csrc"), "-x", "c++",
                               os.path.join(ROOT, "sp1_amd", "csrc", "host_par.hip"), os.path.join(ROOT, "tests", "native", "host_par_stress.cpp"),
                               "-o", exe])
        out = subprocess.run([exe], env=dict(os.environ, SP1HIP_HOST_THREADS=threads), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "threads %s" % threads in out.stdout
    assert "nested: %s 1" % threads in out.stdout
    assert "bad 0" in out.stdout
