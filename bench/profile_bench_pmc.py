"""Workload for the PMC passes (bench/pmc_traffic.sh): ONE whole proof of bench.py's workload (the real-chip core shard) exactly as
bench.py times it. bench.py is asked (SP1HIP_BENCH_PMC_MARK=1) to launch a calibration kernel with a known byte count —
monty_convert: 2^28 words read and written, 4 B per lane coalesced — right before its timed loop: the counter tables keep only what
is dispatched after it, so the set-up (trace building, the preprocessed commitment) is not charged to the proof."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SP1HIP_BENCH_PMC_MARK"] = "1"
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-extras", "--no-verify", "--workload", os.environ.get("SP1HIP_PMC_WORKLOAD", "real")]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
