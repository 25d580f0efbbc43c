#!/bin/bash
# one-rank process group (what every rank of an N-GPU run does, minus the wire) against no process group, alternating
OUT=gpurun_out/r05_ab_dist1.txt
: > $OUT
for rep in 1 2 3; do
  for v in plain dist; do
    if [ $v = dist ]; then export CPC_BENCH_FORCE_DIST=1; else unset CPC_BENCH_FORCE_DIST; fi
    timeout -k 10 200 python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 --no-prof > gpurun_out/ab_d1.json 2>gpurun_out/ab_d1.err || { echo "$v FAILED" >> $OUT; tail -3 gpurun_out/ab_d1.err >> $OUT; continue; }
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_d1.json").read().strip().splitlines()[-1])
print("$v rep $rep: %.3f ms/step exposed %s" % (d["ms_per_step"], d["comm"].get("exposed_ms_per_step")))
PY
  done
done
unset CPC_BENCH_FORCE_DIST
cat $OUT
