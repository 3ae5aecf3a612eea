"""GPU (-m gpu): the RCCL branch of the multi-GPU helpers runs on real device tensors before the driver's 8-GPU node is the
first to execute it. World size 1 on cuda:0 — `torch.distributed` backend "nccl" IS RCCL on ROCm:
communicator creation, all_reduce / all_gather on device tensors (`max_over_ranks`, `gather_blobs`), the compress tree
(`reduce_tree`) and bench.py's `--force-dist --backend nccl` line, which shares its barrier / max-over-ranks code with N = 8.
Each case runs in a child process under a timeout so a rendezvous problem fails the test instead of hanging it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from sp1_amd import shards
assert dist.get_backend() == "nccl" and shards._device().type == "cuda"
assert shards.max_over_ranks(3.25) == 3.25
blobs = {0: b"alpha" * 1000, 3: b"", 7: bytes(range(256)) * 5000}           # 1.28 MB: the size of a shard proof
got = shards.gather_blobs(blobs)
assert got == blobs, sorted(got)
leaves = {i: bytes([i]) * (i + 1) for i in range(5)}
root = shards.reduce_tree(leaves, 5, lambda kids: b"(" + b"|".join(kids) + b")", arity=2)
assert root == b"(((\x00|\x01\x01)|(\x02\x02\x02|\x03\x03\x03\x03))|\x04\x04\x04\x04\x04)", root
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_world1_collectives_on_device_tensors():
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_line_through_rccl_at_one_gpu():
    """`bench.py --gpus 1 --force-dist --backend nccl` at 1/256 of CORE: the driver's N = 1 line can share the N = 8 code path
    (process-group init with device_id, barrier, max-over-ranks on a device tensor, destroy); the line carries
    `verified`, `host_threads` and `dist`."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--scale-log2", "4",
                        "--no-extras", "--force-dist", "--backend", "nccl"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, "bench.py must print exactly one line on stdout:\n" + r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["verified"] is True
    assert line["dist"] == {"initialised": True, "backend": "nccl"}
    assert line["host_threads"] >= 1 and line["value"] > 0
