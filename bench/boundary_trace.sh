#!/bin/bash
# What happens between two consecutive proofs (run on the GPU box): kernels and copies from the query phase of one proof
# to the first encode pass of the next, with the idle gap before each. usage: bench/boundary_trace.sh <out-file>
out=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_b -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras --no-verify > /dev/null 2>&1
python - "$out" <<PY
import csv, glob, sys
ev = []
for fn in glob.glob("/tmp/prof_b/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for fn in glob.glob("/tmp/prof_b/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
idx = [i for i, e in enumerate(ev) if "open_fold_rounds" in e[2]]
i0 = idx[-2] - 3
with open(sys.argv[1], "w") as o:
    end = ev[i0][0]
    for s, e, n in ev[i0:i0 + 70]:
        o.write("%9.1f us  +%7.1f  dur %8.1f  %s\n" % ((s - ev[i0][0]) / 1e3, max(0, s - end) / 1e3, (e - s) / 1e3, n))
        end = max(end, e)
PY
head -60 $out
