#!/usr/bin/env bash
# FETCH_SIZE pass of the bench (its own run, counters only): tools/fetch_pass.sh <tag> [env assignments ...]
set -u
OUT=gpurun_out/fetch_$1
shift
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --no-prof --also "" > $OUT/log.txt 2>&1 || exit 1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        d[r["Kernel_Name"].replace("cpc::", "").replace("void ", "").split("(")[0][:48]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 1e4:
        print(f"{k:48s} launches {len(v):4d}  2xFETCH per launch {2 * 1024 * sum(v) / len(v) / 1e6:9.1f} MB")
PY
