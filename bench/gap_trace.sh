#!/bin/bash
# Where is the GPU idle inside a proof? Kernel trace of `bench.py --steps 3 --warmup 1 --no-extras`; for the last proof:
# busy time (union of kernel intervals), and the idle gaps attributed to the kernel they FOLLOW (a gap after kernel X =
# the host was working / waiting on X's result). usage: bench/gap_trace.sh <out-file>
out=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gap
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2>&1
python - "$out" <<PY
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_gap/*kernel_trace.csv")[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
meta = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])): (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r.get("Queue_Id", "?")) for r in rows}
# proofs start with the first ntt pass after a jagged/basefold tail: find starts of 'first_layer_kernel' and cut one proof = [previous commit start, next commit start)
starts = [i for i, e in enumerate(ev) if "first_layer_kernel" in e[2]]
# the last proof's kernels: from the first kernel after the previous proof's last kernel... approximate: between midpoint marks
last = starts[-1]
prev = starts[-2]
# find the commit start of the last proof: first ntt_fast_pass after prev first_layer
i0 = next(i for i in range(prev + 1, last) if "ntt_fast_pass" in ev[i][2] and ev[i][0] - ev[i - 1][1] > 0 and all("ntt" not in ev[j][2] and "leaf_hash" not in ev[j][2] for j in range(max(prev + 1, i - 3), i)))
seg = ev[i0:]
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy, cur_end, gaps = 0, seg[0][0], collections.defaultdict(float)
gap_n = collections.Counter()
prev_name = None
for s, e, n in seg:
    if s > cur_end:
        if prev_name is not None:
            key = prev_name + "  ->  " + n if "mailbox" in prev_name or "copyBuffer" in prev_name else prev_name
            gaps[key] += (s - cur_end) / 1e3; gap_n[key] += 1
        cur_end = s
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
        prev_name = n
with open(sys.argv[1], "w") as o:
    o.write("last proof: wall %.2f ms, GPU busy (union of kernels) %.2f ms, idle %.2f ms in %d gaps\\n" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, sum(gap_n.values())))
    o.write("idle attributed to the kernel the gap follows (us total | gaps | avg us):\\n")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:30]:
        o.write("%10.1f | %4d | %7.1f | %s\\n" % (v, gap_n[k], v / gap_n[k], k))
    # the largest single gaps, with where in the proof they are and the kernels either side
    singles, cur_end, idx_end = [], seg[0][0], 0
    for i, (s, e, n) in enumerate(seg):
        if s > cur_end and i: singles.append((s - cur_end, cur_end - t0, idx_end, i))
        if e > cur_end: cur_end, idx_end = e, i
    o.write("largest single gaps (us | at ms | kernels before -> kernels after):\\n")
    for g, at, a, b in sorted(singles, reverse=True)[:40]:
        o.write("%8.1f | %7.2f | %s  ->  %s\\n" % (g / 1e3, at / 1e6, " ; ".join(x[2][:40] for x in seg[max(0, a - 1):a + 1]), " ; ".join(x[2][:40] for x in seg[b:b + 2])))
# every launch of the last proof: start offset (us) | idle before it | duration | workgroups | queue | kernel
with open(sys.argv[1].replace(".txt", "") + "_timeline.txt", "w") as o:
    end = t0
    for s, e, n in seg:
        wgs, q = meta.get((s, e), (0, "?"))
        o.write("%9.1f +%7.1f dur %8.1f wgs %6d q%s %s\\n" % ((s - t0) / 1e3, max(0, s - end) / 1e3, (e - s) / 1e3, wgs, q, n.replace("sp1hip::", "").replace("void ", "")))
        end = max(end, e)
PY
cat $out
