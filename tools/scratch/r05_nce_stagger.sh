#!/bin/bash
OUT=gpurun_out/r05_nce_stagger.txt
: > $OUT
for v in base st8 st16 st32 st64 base; do
  if [ $v = base ]; then unset CPC2_HIP_LIB; else export CPC2_HIP_LIB=$PWD/probes/nce_$v.so; fi
  timeout -k 10 200 python bench.py --config small --cpu-seconds 0 --also "" --steps 12 --warmup 4 > gpurun_out/nce_$v.json 2>gpurun_out/nce_$v.err || { echo "$v FAILED" >> $OUT; tail -3 gpurun_out/nce_$v.err >> $OUT; continue; }
  python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/nce_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("%-16s step %.3f ms  infonce_fwd %.1f us  loss %s" % ("$v", d["ms_per_step"], 1e3*k["infonce_fwd"]["ms_per_step"], d["config"]["final_losses"][:2]))
PY
done
cat $OUT
