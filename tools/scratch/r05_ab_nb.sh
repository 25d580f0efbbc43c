#!/bin/bash
# windows per group of the cooperative recurrent kernels at b = 64 (context network on 64 windows): fewer, larger groups leave CUs
# to the work that runs beside the recurrent backward (the criterion's dz sum, the recurrent layer's weight-gradient products)
OUT=gpurun_out/r05_ab_nb.txt
: > $OUT
for cfg in small large; do
for rep in 1 2; do
  for nb in 1 2 4 8; do
    export CPC_COOP_NB_MIN=$nb
    timeout -k 10 200 python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/ab_nb.json 2>gpurun_out/ab_nb.err || { echo "$cfg nb $nb FAILED" >> $OUT; tail -3 gpurun_out/ab_nb.err >> $OUT; continue; }
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_nb.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$cfg nb_min $nb rep $rep: %.3f ms/step" % d["ms_per_step"], {n: round(v["ms_per_step"], 3) for n, v in k.items() if n in ("gru_fwd", "gru_bwd", "gemm_tn", "infonce_bwd")})
PY
  done
done
done
cat $OUT
