#!/bin/bash
# Round-2 closing run (on the GPU box): the whole GPU suite, smoke(), the default bench line, the kernel-stats profile of
# the bench command, the recursion-shard bench. Results under gpurun_out/r02/.
out=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/final_pytest_gpu.txt 2>&1; tail -3 $out/final_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/final_smoke.txt 2>&1; tail -1 $out/final_smoke.txt
timeout 600 python bench.py > $out/final_bench.json 2> $out/final_bench.err; cut -c1-300 $out/final_bench.json
timeout 300 python bench/bench_recursion.py --repeat 5 > $out/final_bench_recursion.txt 2>&1; tail -2 $out/final_bench_recursion.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $out/final_bench_profiled.json 2>/dev/null
cp /tmp/prof_stats/*kernel_stats.csv $out/final_bench_kernel_stats.csv 2>/dev/null
head -12 $out/final_bench_kernel_stats.csv | cut -c1-160
