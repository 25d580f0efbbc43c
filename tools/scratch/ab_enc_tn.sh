#!/bin/bash
# conv3 / conv4 weight-gradient products on the side stream (deferred form) against on the caller's stream (CPC_ENC_TN_MAIN=1)
OUT=gpurun_out/ab_enc_tn.txt
: > $OUT
CPC_SKIP_DP_JOBS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or recurrent_weight or reproducible or deferred" > gpurun_out/ab_enc_tn_tests.log 2>&1; tail -1 gpurun_out/ab_enc_tn_tests.log >> $OUT
for cfg in small large; do
for rep in 1 2 3; do
  for v in main side; do
    unset CPC_ENC_TN_MAIN
    [ $v = main ] && export CPC_ENC_TN_MAIN=1
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
