#!/bin/bash
# Round-6 profile set of the default workload (a full core shard of the reference's fibonacci guest), run on the GPU box through
# gpurun: the PMC traffic table first (bench.py reads it and checks that it belongs to this build's proof shape), then the bench
# line with every extra, then the kernel-trace summary and the per-launch traces of the three sumcheck stages.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06
mkdir -p $out
bash bench/pmc_traffic.sh $GRAFT_REPO_ROOT/$out/r06_traffic_fibonacci.json fibonacci > $out/pmc.log 2>&1
cp $out/r06_traffic_fibonacci.json profiles/r06_traffic_fibonacci.json
timeout 1500 python bench.py --steps 12 --warmup 3 > $out/r06_bench_fibonacci.json 2> $out/r06_bench_fibonacci.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o fib -- python bench.py --no-extras --no-verify --steps 5 --warmup 1 > $out/prof.log 2>&1
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/r06_bench_fibonacci_kernel_stats.csv
bash bench/gkr_trace.sh $GRAFT_REPO_ROOT/$out/r06_gkr_launch_trace.txt > /dev/null 2>&1
bash bench/stage_trace.sh $GRAFT_REPO_ROOT/$out/r06_zerocheck_launch_trace.txt open_sum_kernel zc_gather > /dev/null 2>&1
bash bench/gap_trace.sh $GRAFT_REPO_ROOT/$out/r06_gap_trace.txt > /dev/null 2>&1
python - <<PY
import json
d = json.loads(open("$out/r06_bench_fibonacci.json").read().strip().splitlines()[-1])
st = d["roofline"]["stages"]
print(round(d["ms_per_step"], 2), d["value"], d["unit"], d["verified"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"]["traffic"], d["roofline"]["stale"], d["roofline"]["stale_because"], d["host_cpu_ms_per_proof"])
print({k: round(v["ms"], 2) for k, v in st.items() if isinstance(v, dict) and "ms" in v}, {k: round(v["ms"], 2) for k, v in st["windows"].items()})
if d.get("cpu_baseline"): print(d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"].get("quarter_sample"), d["cpu_baseline"].get("stage_seconds"))
if d.get("in_flight"): print({k: round(v["ms_per_proof"], 1) for k, v in d["in_flight"]["slots"].items()}, (d["in_flight"].get("staged_events") or {}).get("ms_per_proof"), d["in_flight"]["staged_from_host"]["ms_per_proof"])
if d.get("program_run"): print({k: v for k, v in d["program_run"].items() if k not in ("per_shard", "note")})
PY
head -25 $out/r06_bench_fibonacci_kernel_stats.csv | cut -c1-160
