#!/bin/bash
# In-call A/B of two builds of libsp1hip.so on the same GPU box: sp1_amd/lib/prev.so vs the current one.
L=sp1_amd/lib
cp $L/libsp1hip.so $L/cur.so
for rep in 1 2; do
  for v in cur prev; do
    cp $L/$v.so $L/libsp1hip.so
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['ms_per_step'],3), 'iso', r['isolated']['ms_per_step'], 'leaf', r['isolated']['leaf_hash_ms_per_step'], 'ntt', r['isolated']['rs_encode_ms_per_step'])"
  done
done
cp $L/cur.so $L/libsp1hip.so
