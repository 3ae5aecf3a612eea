#!/bin/bash
# per-launch durations of the zerocheck round kernels for one recursion shard proof (run on the GPU box)
# usage: bench/zc_trace.sh <out-file> [script relative to the repo root, default bench/bench_recursion.py --repeat 1] [args]
out=$1; shift
if [ $# -eq 0 ]; then set -- bench/bench_recursion.py --repeat 1; fi
script=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_zc
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_zc -o rec -- python $GRAFT_REPO_ROOT/$script "$@" > /dev/null 2>&1
python - "$out" $ZC_TRACE_ALSO <<PY
import csv, glob, sys
f = glob.glob("/tmp/prof_zc/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = {}
with open(sys.argv[1], "w") as o:
    for r in rows:
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = n.split("(")[0][:70]
        tot[key] = tot.get(key, 0) + d
        if "zc_round" in n or ("gkr::" in n and sys.argv[2:] == ["gkr"]):
            o.write("%s wgs=%d wg=%s dur_us=%.1f\n" % (key, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Workgroup_Size_X"], d))
    o.write("---- totals (us)\n")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
        o.write("%10.1f %s\n" % (v, k))
PY
