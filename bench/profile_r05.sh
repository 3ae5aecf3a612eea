#!/bin/bash
# Round-5 profile set (run on the GPU box through gpurun): PMC traffic tables first (bench.py reads them), then the bench lines
# and the kernel-trace summaries the DESIGN cites. Outputs under gpurun_out/final/ (copied into profiles/ afterwards).
cd $GRAFT_REPO_ROOT
out=gpurun_out/final
mkdir -p $out
bash bench/pmc_traffic.sh $GRAFT_REPO_ROOT/$out/r05_traffic.json real > $out/pmc_real.log 2>&1
cp $out/r05_traffic.json profiles/r05_traffic.json
bash bench/pmc_traffic.sh $GRAFT_REPO_ROOT/$out/r05_traffic_precompile.json precompile > $out/pmc_precompile.log 2>&1
cp $out/r05_traffic_precompile.json profiles/r05_traffic_precompile.json
timeout 900 python bench.py --steps 20 --warmup 5 > $out/r05_bench.json 2> $out/r05_bench.err; echo "bench rc=$?"
timeout 600 python bench.py --workload precompile --no-extras --steps 12 --warmup 3 > $out/r05_bench_precompile.json 2> $out/r05_bench_precompile.err; echo "precompile rc=$?"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_real -o real -- python bench.py --no-extras --no-verify --steps 5 --warmup 1 > $out/prof_real.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_pre -o pre -- python bench.py --workload precompile --no-extras --no-verify --steps 5 --warmup 1 > $out/prof_pre.log 2>&1
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
cp $(find $out/prof_real -name "*kernel_stats.csv" | head -1) $out/r05_bench_kernel_stats.csv
cp $(find $out/prof_pre -name "*kernel_stats.csv" | head -1) $out/r05_bench_precompile_kernel_stats.csv
python - <<PY
import json
for f in ("r05_bench.json", "r05_bench_precompile.json"):
    try:
        d = json.loads(open("$out/" + f).read().strip().splitlines()[-1])
        st = d["roofline"]["stages"]
        print(f, round(d["ms_per_step"], 2), d["value"], d["verified"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["host_cpu_ms_per_proof"], d.get("host_cpu_untimed"))
        print({k: round(v["ms"], 2) for k, v in st.items() if isinstance(v, dict) and "ms" in v}, {k: round(v["ms"], 2) for k, v in st["windows"].items()})
        if d.get("cpu_baseline"): print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("quarter_sample"), d["cpu_baseline"].get("stage_seconds"))
        if d.get("in_flight"): print({k: round(v["ms_per_proof"], 1) for k, v in d["in_flight"]["slots"].items()}, d["in_flight"].get("staged_from_host", {}).get("ms_per_proof"), d["in_flight"].get("staged_global_on_device"))
    except Exception as e:
        print(f, "FAILED", e)
PY
