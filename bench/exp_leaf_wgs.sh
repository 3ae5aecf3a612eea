#!/bin/bash
# Commit-stage overlap experiment (GPU box): whole-proof time and commit-only time against the size of the leaf hash's grid
# (SP1HIP_LEAF_WGS; 0 = one workgroup per 256 rows). usage: bench/exp_leaf_wgs.sh
cd $GRAFT_REPO_ROOT
for w in 0 768 1024 1280 1536 2048; do
  echo -n "SP1HIP_LEAF_WGS=$w: "
  SP1HIP_LEAF_WGS=$w python bench.py --steps 10 --warmup 3 --no-extras --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.2f  leaf_hash %.2f  rs_encode %.2f' % (d['ms_per_step'], d['roofline']['stages']['leaf_hash']['ms'], d['roofline']['stages']['rs_encode']['ms']))"
done
