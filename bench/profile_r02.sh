#!/bin/bash
# Round-2 profiles (run on the GPU box; results under gpurun_out/r02/, the summaries are copied to profiles/):
#  1. rocprofv3 --kernel-trace --stats of the bench command       -> bench_kernel_stats.csv + bench_profiled.json
#  2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes          -> pmc_traffic.txt (per kernel, summed over launches)
out=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-extras > $out/bench_profiled.json 2>/dev/null
cp /tmp/prof_stats/*kernel_stats.csv $out/bench_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/bench/profile_bench_pmc.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/bench/profile_bench_pmc.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
def load(d):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for fn in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0][:80]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    return agg, cnt
f, fc = load("/tmp/pmc_f")
w, wc = load("/tmp/pmc_w")
with open("$out/pmc_traffic.txt", "w") as o:
    o.write("kernel | launches | FETCH_SIZE raw | WRITE_SIZE raw   (rocprofv3 units as reported; calibrate on monty_convert_kernel: 2^28 words = 1048576 KiB each way)\n")
    for k in sorted(f, key=lambda k: -f[k])[:40]:
        o.write("%s | %d | %.1f | %.1f\n" % (k, fc[k], f[k], w.get(k, 0.0)))
PY
head -30 $out/pmc_traffic.txt
head -25 $out/bench_kernel_stats.csv | cut -c1-150
