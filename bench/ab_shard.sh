#!/bin/bash
# In-call A/B of two builds of libsp1hip.so on bench/bench_shard.py (same GPU box): sp1_amd/lib/prev.so vs current.
L=sp1_amd/lib
cp $L/libsp1hip.so $L/cur.so
for rep in 1 2; do
  for v in cur prev; do
    cp $L/$v.so $L/libsp1hip.so
    python bench/bench_shard.py 2>/dev/null | python -c "
import json,sys
ls=[json.loads(l) for l in sys.stdin.read().strip().splitlines()]
print('$v', 'prove_shard_ms', [l['prove_shard_ms'] for l in ls if 'prove_shard_ms' in l], ls[-1].get('stages_ms'))"
  done
done
cp $L/cur.so $L/libsp1hip.so
