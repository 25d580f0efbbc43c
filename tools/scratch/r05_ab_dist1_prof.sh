#!/bin/bash
OUT=gpurun_out/r05_ab_dist1_prof.txt
: > $OUT
for rep in 1 2; do
  for v in "plain:" "plain:--no-prof" "dist:" "dist:--no-prof"; do
    mode=${v%%:*}; flag=${v#*:}
    if [ $mode = dist ]; then export CPC_BENCH_FORCE_DIST=1; else unset CPC_BENCH_FORCE_DIST; fi
    timeout -k 10 200 python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 $flag > gpurun_out/ab_d1.json 2>gpurun_out/ab_d1.err || { echo "$v FAILED" >> $OUT; continue; }
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_d1.json").read().strip().splitlines()[-1])
print("$mode '$flag' rep $rep: %.3f ms/step" % d["ms_per_step"])
PY
  done
done
cat $OUT
