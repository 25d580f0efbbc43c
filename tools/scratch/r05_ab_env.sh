#!/bin/bash
# generic A/B of an environment switch inside bench.py: tools/scratch/r05_ab_env.sh <VAR> <configs> [reps]
VAR=$1; CFGS=${2:-small}; REPS=${3:-3}
OUT=gpurun_out/r05_ab_$VAR.txt
: > $OUT
for cfg in $CFGS; do
for rep in $(seq 1 $REPS); do
  for v in off on; do
    if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
    timeout -k 10 200 python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/ab_env.json 2>gpurun_out/ab_env.err || { echo "$cfg $VAR=$v FAILED" >> $OUT; tail -3 gpurun_out/ab_env.err >> $OUT; continue; }
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_env.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$cfg $VAR=$v rep $rep: %.3f ms/step frac %.4f" % (d["ms_per_step"], d["roofline"]["frac"]), {n: round(v["ms_per_step"], 3) for n, v in k.items() if n in ("conv0_fwd", "conv0_bwd", "gemm_planes_nt", "gemm_planes_tn")}, "loss", d["config"]["final_losses"][:2])
PY
  done
done
done
unset $VAR
cat $OUT
