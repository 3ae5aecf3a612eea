#!/bin/bash
# round 5, last GPU calls (c): the rematerialising zerocheck schedule on the FieldOpCols chips (parity, then the whole rsp block again,
# shard by shard and with three proofs in flight)
mkdir -p gpurun_out/final
timeout 170 python -m pytest tests/test_gpu_riscv_exec.py -x -q -k "big_integer" > gpurun_out/final/pytest_bigint2.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/final/pytest_bigint2.txt
timeout 330 python bench/prove_program.py --program rsp --verify --in-flight 3 --out gpurun_out/final/rsp_whole2.json > /dev/null 2> gpurun_out/final/rsp_whole2.err
echo "rsp rc=$?"
tail -c 400 gpurun_out/final/rsp_whole2.err
