#!/bin/bash
# HBM traffic per kernel of ONE whole core-shaped proof (run on the GPU box): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# --pmc passes (they do not fit one pass: MI355X_MICROARCH.md, TCC counters) over bench/profile_bench_pmc.py, which runs a
# calibration kernel with a known byte count first (monty_convert: 2^28 words read and written = 1 GiB each way).
# Corrections, as the guide's HBM section prescribes and the calibration kernel confirms: rocprofv3 reports both counters in
# KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads (x 2), WRITE_SIZE is exact (x 1).
# A third pass collects SQ_INSTS_VALU (wave-level VALU instructions; x 64 = lane-instructions) and SQ_INSTS_SALU per kernel: the
# numerators of bench.py's VALU roofline fractions (from THIS round's counters, not a constant).
# The shape of the proof the counters belong to (workload, trace cells, first-layer entries, launches per kernel group as the
# library's own timers count them) is written into the table: bench.py holds its live run against it and says `stale` otherwise.
# usage: bench/pmc_traffic.sh <out.json> [fibonacci|real|precompile]     (writes the table bench.py reads: profiles/r06_traffic_<workload>.json)
out=$1
export SP1HIP_PMC_WORKLOAD=${2:-real}
export SP1HIP_BENCH_PMC_META=/tmp/pmc_bench_meta.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_v
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/bench/profile_bench_pmc.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/bench/profile_bench_pmc.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pmc_v -o v -- python $GRAFT_REPO_ROOT/bench/profile_bench_pmc.py > /dev/null 2>&1
python - "$out" <<PY
import csv, glob, json, sys, collections
def load(d, counter=None):
    # only what is dispatched from the calibration kernel on (bench.py launches it right before its timed loop): the set-up —
    # torch's trace building, the preprocessed commitment — is not the proof's
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(fn)) if counter is None or r["Counter_Name"] == counter]
        cal_ids = [int(r["Dispatch_Id"]) for r in rows if "monty_convert_kernel" in r["Kernel_Name"]]
        first = min(cal_ids) if cal_ids else 0
        for r in rows:
            if int(r["Dispatch_Id"]) < first:
                continue
            k = r["Kernel_Name"].split("(")[0]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    return agg, cnt
f, fc = load("/tmp/pmc_f")
w, wc = load("/tmp/pmc_w")
valu, _ = load("/tmp/pmc_v", "SQ_INSTS_VALU")
salu, _ = load("/tmp/pmc_v", "SQ_INSTS_SALU")
cal = next(k for k in f if "monty_convert_kernel" in k)
cal_bytes = (1 << 28) * 4
fetch_scale = cal_bytes / (f[cal] * 1024.0)          # expected 2.0
write_scale = cal_bytes / (w[cal] * 1024.0)          # expected 1.0
groups = {   # bench.py's kernel names -> substrings of the HIP kernel names
    "leaf_hash": ["leaf_hash_part_kernel", "leaf_hash_kernel"],
    "rs_encode": ["ntt_fast_pass"],
    "zerocheck_round": ["zc_round_kernel", "zc_macro_kernel", "zc_biv_round_kernel", "zc_biv_macro_kernel", "zc_biv_keccak_kernel", "zc_biv_corner_kernel", "zc_poly_kernel", "zc_biv_poly_kernel", "zc_keccak3_kernel"],
    "zerocheck_fix": ["zc_fix_kernel", "zc_fix2_kernel"],
    "gkr_pass": ["gkr_pass"],
    "compress": ["compress_layer", "compress_top"],
    "gkr_first_layer": ["first_layer_kernel", "first_layers_kernel"],
    "gkr_transition": ["transition_kernel", "transition2_kernel"],
    "jagged_fold": ["jg_fold"],
}
kernels = {}
for name, subs in groups.items():
    ks = [k for k in f if any(s in k for s in subs)]
    if not ks:
        continue
    fetch = sum(f[k] for k in ks) * 1024 * 2.0       # the guide's gfx950 correction (the calibration kernel reports its own measured scale below)
    write = sum(w.get(k, 0.0) for k in ks) * 1024 * 1.0
    launches = sum(fc[k] for k in ks)
    kernels[name] = {"launches_per_proof": launches, "fetch_bytes_per_proof": fetch, "write_bytes_per_proof": write,
                     "hbm_bytes_per_proof": fetch + write, "hbm_bytes_per_launch": (fetch + write) / launches,
                     "valu_wave_insts_per_proof": sum(valu.get(k, 0.0) for k in ks),
                     "valu_lane_insts_per_proof": 64.0 * sum(valu.get(k, 0.0) for k in ks),
                     "salu_insts_per_proof": sum(salu.get(k, 0.0) for k in ks)}
# every kernel of the library inside the proof (the calibration kernel and torch's trace-building kernels are not the proof's)
own = [k for k in f if ("sp1hip::" in k or "rocclr" in k) and "monty_convert" not in k and "tracegen" not in k]
kernels["all_kernels"] = {"launches_per_proof": sum(fc[k] for k in own),
                          "fetch_bytes_per_proof": sum(f[k] for k in own) * 2048.0, "write_bytes_per_proof": sum(w.get(k, 0.0) for k in own) * 1024.0,
                          "hbm_bytes_per_proof": sum(f[k] for k in own) * 2048.0 + sum(w.get(k, 0.0) for k in own) * 1024.0,
                          "valu_wave_insts_per_proof": sum(valu.get(k, 0.0) for k in own),
                          "valu_lane_insts_per_proof": 64.0 * sum(valu.get(k, 0.0) for k in own),
                          "salu_insts_per_proof": sum(salu.get(k, 0.0) for k in own)}
kernels["all_kernels"]["hbm_bytes_per_launch"] = kernels["all_kernels"]["hbm_bytes_per_proof"] / max(1, kernels["all_kernels"]["launches_per_proof"])
import os
try:
    shape = json.load(open(os.environ["SP1HIP_BENCH_PMC_META"]))
except (OSError, KeyError, ValueError):
    shape = None
json.dump({"workload": os.environ.get("SP1HIP_PMC_WORKLOAD", "real"), "shape": shape,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU SQ_INSTS_SALU, three separate passes over one proof of the "
                     "bench workload named in `workload` (bench/pmc_traffic.sh -> profiles/r06_traffic*.json); KiB units, FETCH x 2 (gfx950: 64 B "
                     "tallied per 128 B request), WRITE x 1; SQ_INSTS_VALU counts wave instructions (x 64 lanes)",
           "calibration": {"kernel": "monty_convert_kernel, 2^28 words each way", "fetch_scale_measured": fetch_scale,
                           "write_scale_measured": write_scale},
           "kernels": kernels}, open(sys.argv[1], "w"), indent=1)
print(json.dumps(kernels, indent=1)[:3000])
PY
