#!/bin/bash
# output stores of the split-in-kernel NT family: four-byte (CPC_GEMM_SCALAR_OUT=1) / 16-byte through LDS / the same non-temporal
OUT=gpurun_out/ab_x6_out.txt
: > $OUT
for cfg in small transformer large; do
for rep in 1 2; do
  for v in scalar staged nt; do
    unset CPC_GEMM_SCALAR_OUT CPC_GEMM_NT_OUT
    [ $v = scalar ] && export CPC_GEMM_SCALAR_OUT=1
    [ $v = nt ] && export CPC_GEMM_NT_OUT=1
    timeout -k 10 200 python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/ab_x_$v.json 2>gpurun_out/ab_x_$v.err || tail -5 gpurun_out/ab_x_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_x_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$cfg $v rep $rep: %.3f ms/step" % d["ms_per_step"], {n: (round(v["ms_per_step"], 3), v["launches_per_step"]) for n, v in k.items() if n in ("gemm_nt", "gemm_tn")})
PY
  done
done
done
cat $OUT
