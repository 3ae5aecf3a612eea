#!/bin/bash
# runs bench/ubench_mfma on the GPU box with the SQ counters north_star asks for; writes gpurun_out/r02/ubench_mfma*.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r02
$GRAFT_REPO_ROOT/bench/ubench_mfma > $GRAFT_REPO_ROOT/gpurun_out/r02/ubench_mfma.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_mfma -o m -- $GRAFT_REPO_ROOT/bench/ubench_mfma 1048576 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("/tmp/pmc_mfma/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$GRAFT_REPO_ROOT/gpurun_out/r02/ubench_mfma_pmc.txt", "w") as o:
    for k, v in agg.items():
        if "ext_" in k:
            o.write(k + " " + " ".join("%s=%d" % kv for kv in sorted(v.items())) + "\n")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r02/ubench_mfma.txt $GRAFT_REPO_ROOT/gpurun_out/r02/ubench_mfma_pmc.txt
