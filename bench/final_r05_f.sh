#!/bin/bash
# round 5, last GPU call (f): the whole rsp block, direct pass + two passes with three proofs in flight (cold / warm slot arenas)
mkdir -p gpurun_out/final
timeout 140 python bench/prove_program.py --program rsp --in-flight 3 --out gpurun_out/final/rsp_whole5.json > /dev/null 2> gpurun_out/final/rsp_whole5.err
echo "rsp rc=$?"
tail -c 200 gpurun_out/final/rsp_whole5.err
