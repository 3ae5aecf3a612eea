#!/bin/bash
# round 5, last GPU calls (d): the whole rsp block, shard by shard and through the prover pool (three proofs in flight)
mkdir -p gpurun_out/final
timeout 330 python bench/prove_program.py --program rsp --verify --in-flight 3 --out gpurun_out/final/rsp_whole3.json > /dev/null 2> gpurun_out/final/rsp_whole3.err
echo "rsp rc=$?"
tail -c 600 gpurun_out/final/rsp_whole3.err
