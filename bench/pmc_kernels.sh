#!/bin/bash
# SQ counters of selected kernels for one run of a script (run on the GPU box).
# usage: bench/pmc_kernels.sh <out-file> <kernel-name-substring> "<counters...>" <script> [args]
out=$1; pat=$2; counters=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_k
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d /tmp/pmc_k -o k -- python $GRAFT_REPO_ROOT/"$@" > /dev/null 2>&1
python - "$out" "$pat" <<PY
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for fn in glob.glob("/tmp/pmc_k/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if sys.argv[2] in r["Kernel_Name"]:
            key = r["Kernel_Name"].split("(")[0][:70] + " wg=" + r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(key, r["Counter_Name"])] += 1
with open(sys.argv[1], "a") as o:
    for k, v in agg.items():
        o.write(k + " launches=%d " % max(cnt[(k, c)] for c in v) + " ".join("%s=%d" % kv for kv in sorted(v.items())) + "\n")
PY
