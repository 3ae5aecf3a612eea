#!/bin/bash
# eight prover processes on ONE node's host — here all eight share the single GPU of the box (gloo for the
# barrier), proving small shards (scale 4^-3) so that the host side (transcript, pollers, helpers, launch calls) is what is
# exercised. Reports per-proof wall and host CPU (user + system of all threads) at world 1 and world 8.
# usage: bench/world8_host.sh <out.txt>       (run on the GPU box)
out=$1
cd $GRAFT_REPO_ROOT
{
echo "# python bench.py --scale-log2 3 --steps 40 --warmup 4 --no-extras --no-verify  (one process)"
python bench.py --scale-log2 3 --steps 40 --warmup 4 --no-extras --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k:d[k] for k in ('n_gpus','ms_per_step','proofs_per_s','host_cpu_ms_per_proof','host_threads')}))"
echo "# the same under torch.distributed.run --nproc-per-node 8 --backend gloo: 8 processes, ONE GPU, $(nproc) CPUs visible, cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --backend gloo --scale-log2 3 --steps 40 --warmup 4 --no-extras --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d2={k:d[k] for k in ('n_gpus','ms_per_step','proofs_per_s','host_cpu_ms_per_proof','host_threads')}; d2['ms_per_proof_aggregate']=d['ms_per_step']/8; print(json.dumps(d2))"
} > $out 2>&1
cat $out
