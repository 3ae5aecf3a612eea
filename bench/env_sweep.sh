#!/bin/bash
# A/B runs of environment knobs on the default bench workload: ms per proof, stage windows, host CPU per proof.
# usage: bench/env_sweep.sh "VAR=1" "OTHER=2 THIRD=x" ...   (each argument is one configuration; "X=1" = the defaults)
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 8 --warmup 2 --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
    return None
print(round(d['ms_per_step'],2), {k:round(v['ms'],2) for k,v in d['roofline']['stages']['windows'].items()}, 'cpu', round(find(d,'host_cpu_ms_per_proof'),1))"
done
