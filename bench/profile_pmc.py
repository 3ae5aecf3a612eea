"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): a calibration kernel with a
known byte count in the same access width as the hot kernels (4 B per lane, coalesced:
monty_convert reads and writes 2^28 words = 1.074 GB each way), then two commit steps of the bench
workload (NTT passes + leaf hash + compress)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sp1_amd import api  # noqa: E402

torch.cuda.set_device(0)
L = api._L()
n = 1 << 28
buf = torch.zeros(n, dtype=torch.int32, device="cuda")
api.check(L.sp1hip_to_monty(api._dptr(buf), n, api._stream_ptr()))
torch.cuda.synchronize()
del buf
g = torch.Generator(device="cuda")
g.manual_seed(42)
mles = [api.ColMajor(torch.randint(0, api.P, (32 << 20,), dtype=torch.int32, device="cuda", generator=g), 1 << 20, 32)
        for _ in range(8)]
prover = api.BasefoldProver(2, 124, 16)
for _ in range(2):
    commit, pd = prover.commit_mles(mles)
    del pd
torch.cuda.synchronize()
print("done", commit[0])
