#!/bin/bash
# Round-4 profile set (run on the GPU box; everything lands under gpurun_out/r04/, the summaries are copied to profiles/):
#  0. FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU+SALU in separate --pmc passes              -> traffic.json  (bench.py reads profiles/r04_traffic.json)
#  1. bench.py as the driver runs it (extras included)                                   -> bench.json
#  2. rocprofv3 --kernel-trace --stats --marker-trace of the bench command (no extras)   -> bench_kernel_stats.csv, marker ranges
#  4. where the GPU idles inside one proof (bench/gap_trace.sh)                          -> gap_trace.txt
#  5. stage walls of the real-chip shard and the recursion shard                         -> real_stages.txt, bench_recursion.txt
out=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $out
cd $GRAFT_REPO_ROOT
# the PMC table first: bench.py reads profiles/r04_traffic.json for the roofline object, and it must be the table of THIS build
timeout 900 bash bench/pmc_traffic.sh $out/traffic.json > $out/traffic.log 2>&1
cp $out/traffic.json profiles/r04_traffic.json
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-extras --no-verify > $out/bench_profiled_run.json 2>/dev/null
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $out/bench_kernel_stats.csv 2>/dev/null
cp $(find /tmp/prof_stats -name '*marker_api_trace.csv' | head -1) $out/marker_trace_full.csv 2>/dev/null
python - $out <<'PY'
import csv, sys, collections, os
p = os.path.join(sys.argv[1], "marker_trace_full.csv")
if os.path.exists(p):
    rows = list(csv.DictReader(open(p)))
    agg = collections.OrderedDict()
    for r in rows:
        name = r.get("Function") or r.get("Name") or r.get("Message") or "?"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += d
    with open(os.path.join(sys.argv[1], "marker_ranges.txt"), "w") as o:
        o.write("# roctx ranges (rocprofv3 --marker-trace) of `bench.py --steps 4 --warmup 1 --no-extras --no-verify`: name | count | total ms | ms per range\n")
        for k, (n, t) in agg.items(): o.write("%-28s %4d %10.3f %9.3f\n" % (k, n, t, t / n))
    os.remove(p)
PY
cd $GRAFT_REPO_ROOT
timeout 300 bash bench/gap_trace.sh $out/gap_trace.txt > /dev/null 2>&1
SP1HIP_SHARD_TIMING=1 timeout 200 python bench/bench_real.py 0 4 > $out/real_stages.txt 2>&1
SP1HIP_SHARD_TIMING=1 timeout 200 python bench/bench_recursion.py --repeat 4 --stages > $out/bench_recursion.txt 2>&1
head -c 400 $out/bench.json; echo; tail -2 $out/bench.err; cat $out/marker_ranges.txt 2>/dev/null | head -20; head -3 $out/gap_trace.txt
