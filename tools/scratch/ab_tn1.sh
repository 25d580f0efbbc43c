#!/bin/bash
# EXPERIMENT: conv1's weight-gradient product beside conv0's backward (CPC_ENC_TN1_SIDE=1) against on the caller's stream
OUT=gpurun_out/ab_tn1.txt
: > $OUT
CPC_ENC_TN1_SIDE=1 CPC_SKIP_DP_JOBS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or recurrent_weight or reproducible" > gpurun_out/ab_tests.log 2>&1; tail -1 gpurun_out/ab_tests.log >> $OUT
for cfg in small large; do
for rep in 1 2 3; do
  for v in main side; do
    unset CPC_ENC_TN1_SIDE
    [ $v = side ] && export CPC_ENC_TN1_SIDE=1
    timeout -k 10 200 python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 40 --warmup 8 > gpurun_out/ab_x_$v.json 2>gpurun_out/ab_x_$v.err || tail -5 gpurun_out/ab_x_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_x_$v.json").read().strip().splitlines()[-1])
print("$cfg $v rep $rep: %.3f ms/step" % d["ms_per_step"], "frac", d["roofline"]["frac"], "loss", d["config"]["final_losses"][:2])
PY
  done
done
done
cat $OUT
