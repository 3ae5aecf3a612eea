#!/bin/bash
# Kernel timeline of the LAST proof of `bench/bench_shard.py --core-shaped --repeat 2` (run on the GPU box): every kernel from
# the first one whose name contains <from-pattern> (last occurrence of a run of them) to the end of the proof, with its
# start offset, duration and the idle gap before it. usage: bench/timeline.sh <out-file> <from-pattern>
out=$1; pat=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o t -- python $GRAFT_REPO_ROOT/bench/bench_shard.py --core-shaped --repeat 2 > /dev/null 2>&1
python - "$out" "$pat" <<PY
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True)[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:64], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
idx = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
# start of the last run of matching kernels: walk back from the last match while matches are < 50 ms apart
i0 = idx[-1]
for i in reversed(idx):
    if ev[i0][0] - ev[i][0] < 50e6: i0 = i
seg = ev[i0:]
t0 = seg[0][0]
with open(sys.argv[1], "w") as o:
    end = t0
    for s, e, n, wgs in seg:
        o.write("%9.1f us  +%7.1f  dur %8.1f  wgs %6d  %s\n" % ((s - t0) / 1e3, max(0, s - end) / 1e3, (e - s) / 1e3, wgs, n))
        end = max(end, e)
    o.write("total %.2f ms, %d kernels\n" % ((end - t0) / 1e6, len(seg)))
PY
tail -1 $out
