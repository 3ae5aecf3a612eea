#!/bin/bash
# per-launch durations of the zerocheck kernels (compiled zc_jit_* and interpreted zc_round_kernel) of one core-shaped proof
# usage: bench/jit_trace.sh <out-file>
out=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_jit
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_jit -o rec -- python $GRAFT_REPO_ROOT/bench/bench_shard.py --core-shaped --repeat 1 > /dev/null 2>&1
python - "$out" <<PY
import csv, glob, sys
f = glob.glob("/tmp/prof_jit/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[1], "w") as o:
    prev_end = None
    for r in rows:
        n = r["Kernel_Name"]
        if "zc_" not in n: prev_end = None; continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        o.write("%-40s wgs=%6d dur_us=%9.1f gap_before_us=%7.1f\n" % (n.split("(")[0][:40], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), (e - s) / 1e3, gap))
        prev_end = e
PY
