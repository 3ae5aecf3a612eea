#!/bin/bash
# round 5, last GPU calls (b): the WHOLE rsp block (every shard, verified per kind), smoke, the default bench line
mkdir -p gpurun_out/final
timeout 330 python bench/prove_program.py --program rsp --verify --out gpurun_out/final/rsp_whole.json > /dev/null 2> gpurun_out/final/rsp_whole.err
echo "rsp rc=$?"
tail -c 600 gpurun_out/final/rsp_whole.err
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/final/smoke.txt
timeout 240 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
echo "bench rc=$?"; head -c 600 gpurun_out/final/bench.json
