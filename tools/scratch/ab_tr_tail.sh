#!/bin/bash
# transformer AR: the layer's parameter-gradient work on the side stream (shipped) against the immediate form (CPC_NO_GRAD_TAIL=1
# switches every deferral off: compare with ab_gru_tail.sh's small numbers for the encoder's share)
OUT=gpurun_out/ab_tr_tail.txt
: > $OUT
for rep in 1 2 3; do
  for v in immediate deferred; do
    unset CPC_NO_GRAD_TAIL
    [ $v = immediate ] && export CPC_NO_GRAD_TAIL=1
    timeout -k 10 200 python bench.py --config transformer --cpu-seconds 0 --also "" --steps 40 --warmup 8 > gpurun_out/ab_x_$v.json 2>gpurun_out/ab_x_$v.err || tail -5 gpurun_out/ab_x_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_x_$v.json").read().strip().splitlines()[-1])
print("transformer $v rep $rep: %.3f ms/step" % d["ms_per_step"], "frac", d["roofline"]["frac"], "loss", d["config"]["final_losses"][:2])
PY
  done
done
cat $OUT
