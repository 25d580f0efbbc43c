#!/bin/bash
# the criterion's dz sum held to 64 registers (can sit beside the matrix-pipe GRU backward: 2 x 222 of 512 registers per SIMD taken)
# against the 68-register kernel (tools/variant: -DNCE_GATHER_WIDE), CPC-small and CPC-large, alternating pairs on one box
OUT=gpurun_out/ab_gather.txt
: > $OUT
for cfg in small large; do
for rep in 1 2 3; do
  for v in narrow wide; do
    if [ $v = wide ]; then export CPC2_HIP_LIB=$PWD/tools/variant/libcpc2_hip.so; else unset CPC2_HIP_LIB; fi
    timeout -k 10 200 python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/ab_g_$v.json 2>gpurun_out/ab_g_$v.err || tail -5 gpurun_out/ab_g_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_g_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$cfg $v rep $rep: %.3f ms/step" % d["ms_per_step"], {n: round(v["ms_per_step"], 3) for n, v in k.items() if "gemm_tn" == n or "gru_bwd" in n or "infonce_bwd" in n})
PY
  done
done
done
cat $OUT
