"""Workload for the round-2 PMC passes: a calibration kernel with a known byte count (monty_convert: 2^28 words read and
written, 4 B per lane coalesced), then ONE whole proof of bench.py's workload (the real-chip core shard) exactly as bench.py times it."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sp1_amd import api  # noqa: E402

torch.cuda.set_device(0)
n = 1 << 28
buf = torch.zeros(n, dtype=torch.int32, device="cuda")
api.check(api._L().sp1hip_to_monty(api._dptr(buf), n, api._stream_ptr()))
torch.cuda.synchronize()
del buf
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-extras", "--no-verify"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
