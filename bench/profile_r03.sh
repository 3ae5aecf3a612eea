#!/bin/bash
# Round-3 profile set (run on the GPU box; everything lands under gpurun_out/r03/, the summaries are copied to profiles/):
#  1. bench.py as the driver runs it (extras included)                        -> bench.json
#  2. rocprofv3 --kernel-trace --stats of the bench command (no extras)       -> bench_kernel_stats.csv + bench_profiled_run.json
#  3. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (bench/pmc_traffic.sh) -> traffic.json
#  4. where the GPU idles inside one proof (bench/gap_trace.sh)               -> gap_trace.txt
#  5. stage walls + kernel timers of the core-shaped shard and the recursion shard
#  6. per-launch durations and SQ counters of the LogUp-GKR pass kernels
out=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-extras --no-verify > $out/bench_profiled_run.json 2>/dev/null
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $out/bench_kernel_stats.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
bash bench/pmc_traffic.sh $out/traffic.json > $out/traffic.log 2>&1
bash bench/gap_trace.sh $out/gap_trace.txt > /dev/null 2>&1
python bench/bench_shard.py --core-shaped --repeat 3 > $out/bench_shard_core_stages.txt 2>&1
python bench/bench_recursion.py --repeat 4 --stages > $out/bench_recursion.txt 2>&1
ZC_TRACE_ALSO=gkr bash bench/zc_trace.sh $out/gkr_zc_launch_trace_core.txt bench/bench_shard.py --core-shaped --repeat 1 > /dev/null 2>&1
bash bench/pmc_kernels.sh $out/pmc_gkr_pass.txt gkr_pass "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" bench/bench_shard.py --core-shaped --repeat 1
bash bench/pmc_kernels.sh $out/pmc_gkr_pass.txt gkr_pass "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" bench/bench_shard.py --core-shaped --repeat 1
head -c 600 $out/bench.json; echo; tail -3 $out/bench.err; cat $out/traffic.log | head -40; cat $out/gap_trace.txt | head -12
