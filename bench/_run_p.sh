mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_jagged.py tests/test_gpu_shard.py tests/test_gpu_core_shard.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02/t_q.txt 2>&1; tail -2 gpurun_out/r02/t_q.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02/bench_q.json 2> gpurun_out/r02/bench_q.err
timeout 300 python bench/bench_shard.py --core-shaped > gpurun_out/r02/shard_core_q.txt 2>&1; tail -1 gpurun_out/r02/shard_core_q.txt | cut -c100-400
