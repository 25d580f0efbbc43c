#!/bin/bash
# issue / wait counters of the similarity kernel (and the kernels around it) inside the CPC-small step: two --pmc passes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_nce
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --config small --steps 3 --warmup 2 --cpu-seconds 0 --no-prof --also="
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.log 2>&1 || echo "p4 failed (counter names?)"
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_nce"
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(root + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("cpc::", "").replace("void ", "")
        if "infonce_fwd" in k or "gemm_planes_kernel<0, false, 6, true" in k:
            a = agg[(k[:40], r["Counter_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
with open(root + "/summary.txt", "w") as out:
    for (k, c), (n, s) in sorted(agg.items()):
        line = "%-42s %-28s mean per dispatch %.4g" % (k, c, s / n)
        print(line); out.write(line + "\n")
PY
