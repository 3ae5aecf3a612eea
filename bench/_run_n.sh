mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_zerocheck.py tests/test_gpu_core_shard.py tests/test_gpu_recursion.py tests/test_gpu_shard.py -m gpu -x -q > gpurun_out/r02/t_n.txt 2>&1; tail -3 gpurun_out/r02/t_n.txt
timeout 200 python bench/bench_recursion.py --repeat 4 > gpurun_out/r02/rec_n.txt 2>&1; tail -1 gpurun_out/r02/rec_n.txt | grep -o '"prove_shard_ms": [0-9.]*\|"zerocheck_round_ms": [0-9.]*' | tr '\n' ' '; echo
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02/bench_n.json 2> gpurun_out/r02/bench_n.err
