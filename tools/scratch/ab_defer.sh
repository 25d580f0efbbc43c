#!/bin/bash
# the criterion's deferred backward (dz sum + predictor dW on the side stream) against the immediate form, CPC-small and CPC-large
OUT=gpurun_out/ab_defer.txt
: > $OUT
for cfg in small large; do
for rep in 1 2; do
  for v in defer nodefer; do
    if [ $v = nodefer ]; then export CPC_NCE_NO_DEFER=1; else unset CPC_NCE_NO_DEFER; fi
    timeout -k 10 200 python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/ab_d_$v.json 2>gpurun_out/ab_d_$v.err || tail -5 gpurun_out/ab_d_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_d_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$cfg $v rep $rep: %.3f ms/step" % d["ms_per_step"], {n: round(v["ms_per_step"], 3) for n, v in k.items() if "gemm_tn" in n or "gemm_nt" == n or "gru" in n or "infonce" in n})
PY
  done
done
done
cat $OUT
