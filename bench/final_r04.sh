#!/bin/bash
# full GPU test suite, smoke(), then the round-4 profile set
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/final/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/final/smoke.log
timeout 1500 bash bench/profile_r04.sh > gpurun_out/final/profile.log 2>&1
tail -3 gpurun_out/final/gpu_tests.log; tail -2 gpurun_out/final/smoke.log; tail -12 gpurun_out/final/profile.log
