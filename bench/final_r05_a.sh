#!/bin/bash
# round 5, last GPU calls (a): parity of the big-integer precompile shards, then a bounded sample of every shard kind of the rsp block
mkdir -p gpurun_out/final
timeout 170 python -m pytest tests/test_gpu_riscv_exec.py -x -q -k "big_integer" > gpurun_out/final/pytest_bigint.txt 2>&1
echo "pytest rc=$?"
tail -3 gpurun_out/final/pytest_bigint.txt
timeout 400 python bench/prove_program.py --program rsp --core-shards 2 --verify --out gpurun_out/final/rsp_kinds.json > /dev/null 2> gpurun_out/final/rsp_kinds.err
echo "rsp rc=$?"
tail -c 1500 gpurun_out/final/rsp_kinds.err
