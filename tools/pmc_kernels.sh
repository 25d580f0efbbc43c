#!/bin/bash
# Counter passes over one bench step set, per-kernel means:  tools/pmc_kernels.sh <tag> "<kernel name substrings, | separated>" [config]
# (three separate --pmc runs; no trace domains beside them; FETCH_SIZE / WRITE_SIZE have passes of their own in tools/gpu_profile.sh:
#  together with SQ counters the run aborts)
set -u
export TMPDIR=/tmp
tag=$1; names=$2; cfg=${3:-small}
out=gpurun_out/pmck_$tag
mkdir -p $out
cmd="python3 bench.py --config $cfg --steps 3 --warmup 2 --cpu-seconds 0 --no-prof --also="
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- $cmd > $out/p1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p2 -- $cmd > $out/p2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d $out/p3 -- $cmd > $out/p3.log 2>&1 || exit 1
python3 - "$out" "$names" <<'PY'
import csv, glob, sys, collections
root, names = sys.argv[1], sys.argv[2].split("|")
agg = collections.defaultdict(list)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("cpc::", "").replace("void ", "").split("(")[0]
        if any(n in k for n in names):
            agg[(k[:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:46s} {c:26s} mean per dispatch {sum(v) / len(v):.4g}   ({len(v)} dispatches)")
PY
