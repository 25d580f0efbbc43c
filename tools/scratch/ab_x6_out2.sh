#!/bin/bash
OUT=gpurun_out/ab_x6_out2.txt
: > $OUT
for rep in 1 2 3; do
  for v in staged staged_slabscalar nt nt_slabscalar; do
    unset CPC_GEMM_SCALAR_OUT CPC_GEMM_NT_OUT CPC_GEMM_SLAB_SCALAR
    case $v in nt*) export CPC_GEMM_NT_OUT=1;; esac
    case $v in *slabscalar) export CPC_GEMM_SLAB_SCALAR=1;; esac
    timeout -k 10 200 python bench.py --config small --cpu-seconds 0 --also "" --steps 40 --warmup 8 > gpurun_out/ab_x_$v.json 2>gpurun_out/ab_x_$v.err || tail -5 gpurun_out/ab_x_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_x_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("small $v rep $rep: %.3f ms/step" % d["ms_per_step"], {n: (round(v["ms_per_step"], 3), v["launches_per_step"]) for n, v in k.items() if n in ("gemm_nt", "gemm_tn")})
PY
  done
done
cat $OUT
