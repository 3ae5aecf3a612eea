#!/bin/bash
# Idle intervals of ONE whole proof (the last of `bench.py --steps 3 --warmup 1 --no-extras --no-verify`): every gap of more than
# <min-us> (default 60) between the end of all earlier kernels and the start of the next one, with the kernels on both sides.
# usage: bench/gap_trace.sh <out-file> [min-us] [workload]
out=$1; minus=${2:-60}; wl=${3:-fibonacci}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gap
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -o g -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-extras --no-verify > /dev/null 2>&1
python - "$out" "$minus" <<PY
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_gap/**/*kernel_trace.csv", recursive=True)[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sp1hip::", "").replace("void ", "")[:60]) for r in rows)
fl = [i for i, e in enumerate(ev) if "first_layers_kernel" in e[2]]
# the last proof: from the first RS-encode pass after the previous proof's first layer to the last kernel of the trace
i0 = next(i for i in range(fl[-2], len(ev)) if "ntt_fast_pass" in ev[i][2] and ev[i][0] > ev[fl[-2]][1] + 40e6)
seg = ev[i0:]
t0 = seg[0][0]
minus = float(sys.argv[2])
with open(sys.argv[1], "w") as o:
    o.write("the last proof: %d launches, %.2f ms from its first to its last kernel\n" % (len(seg), (max(e[1] for e in seg) - t0) / 1e6))
    busy_end, prev, idle_small, idle_big = seg[0][1], seg[0][2], 0.0, 0.0
    o.write("   at ms | idle us | after -> before\n")
    for s, e, n in seg[1:]:
        gap = (s - busy_end) / 1e3
        if gap > minus:
            o.write("%8.3f | %7.1f | %s -> %s\n" % ((s - t0) / 1e6, gap, prev, n)); idle_big += gap
        elif gap > 0:
            idle_small += gap
        if e > busy_end: busy_end, prev = e, n
    o.write("idle in gaps over %.0f us: %.2f ms; in shorter gaps: %.2f ms\n" % (minus, idle_big / 1e3, idle_small / 1e3))
PY
