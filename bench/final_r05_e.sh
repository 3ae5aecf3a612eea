#!/bin/bash
# round 5, last GPU calls (e): near-fit arena; the whole rsp block again (direct + three in flight), then the pool / executor GPU tests
mkdir -p gpurun_out/final
timeout 300 python bench/prove_program.py --program rsp --in-flight 3 --out gpurun_out/final/rsp_whole4.json > /dev/null 2> gpurun_out/final/rsp_whole4.err
echo "rsp rc=$?"
tail -c 300 gpurun_out/final/rsp_whole4.err
timeout 230 python -m pytest tests/ -x -q -m gpu -k "pool or real_program or big_integer or corrupted" > gpurun_out/final/pytest_pool_exec.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/final/pytest_pool_exec.txt
