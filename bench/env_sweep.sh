run() {
  echo "== $1"
  env $1 python bench.py --steps 8 --warmup 2 --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('extras',d)
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
    return None
print(round(d['ms_per_step'],2), 'cpu', round(find(d,'host_cpu_ms_per_proof'),1), find(d,'host_cpu_ms_by_thread'))"
}
run "X=1"
run "HSA_ENABLE_MWAITX=1"
run "GPU_MAX_HW_QUEUES=2"
run "SP1HIP_WAIT=spin"
run "SP1HIP_ZC_FORK=0"
run "HSA_ENABLE_INTERRUPT=0"
run "AMD_DIRECT_DISPATCH=0"
