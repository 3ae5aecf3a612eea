#!/bin/bash
# Every kernel launch of ONE proof between two marker kernels, in order: name, workgroups, duration, idle time before it (kernel
# trace of `bench.py --steps 3 --warmup 1 --no-extras --no-verify`, last proof).
# usage: bench/stage_trace.sh <out-file> <first-kernel-substring> <last-kernel-substring> [workload] [command]
#   command: what to trace instead of bench.py, relative to the repository root (its LAST proof is the one listed), e.g.
#   "bench/prove_program.py --program rsp --only-kinds secp256k1_add"
#   LogUp-GKR: first_layer open_sum_kernel | zerocheck: (the launch after) open_sum_kernel .. zc_gather | whole proof: ntt_fast_pass fold_round
out=$1; first=$2; last=$3
wl=${4:-fibonacci}
cmd=${5:-"bench.py --workload $wl --steps 3 --warmup 1 --no-extras --no-verify"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stage
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_stage -o g -- python $GRAFT_REPO_ROOT/$cmd > /dev/null 2>&1
python - "$out" "$first" "$last" <<PY
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_stage/**/*kernel_trace.csv", recursive=True)[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sp1hip::", "").replace("void ", "")[:70],
             int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]))) for r in rows)
first, last = sys.argv[2], sys.argv[3]
i0 = [i for i, e in enumerate(ev) if first in e[2]][-1]
if first == last:                       # "the stage that starts right after the last launch of this kernel"
    i0 += 1
    i1 = len(ev) - 1
else:
    i1 = [i for i in range(i0, len(ev)) if last in ev[i][2]][-1]
# a stage = a contiguous run: when the first marker is itself repeated inside (ntt passes), start at the first of the last proof
seg = ev[i0:i1 + 1]
t0 = seg[0][0]
with open(sys.argv[1], "w") as o:
    o.write("%d launches, %.2f ms from the first to the last\\n" % (len(seg), (seg[-1][1] - t0) / 1e6))
    o.write("   at ms | dur us | gap us | workgroups | kernel\\n")
    prev_end = t0
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for s, e, n, wg in seg:
        gap = max(0, s - prev_end) / 1e3
        o.write("%8.3f | %7.1f | %6.1f | %9d | %s\\n" % ((s - t0) / 1e6, (e - s) / 1e3, gap, wg, n))
        a = agg[n]; a[0] += 1; a[1] += (e - s) / 1e3; a[2] += gap
        prev_end = max(prev_end, e)
    o.write("\\nby kernel: launches | kernel us | idle before us\\n")
    for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write("%5d | %9.1f | %9.1f | %s\\n" % (c, d, g, n))
    big = sum(e - s for s, e, n, wg in seg if e - s > 100000) / 1e6
    o.write("\\nlaunches over 100 us: %.2f ms in total; the rest: %.2f ms of kernels and %.2f ms of idle\\n"
            % (big, sum(e - s for s, e, n, wg in seg) / 1e6 - big, sum(v[2] for v in agg.values()) / 1e3))
PY
