#!/bin/bash
# conv0_bwd_dma_kernel inside the CPC-small step: probe builds that leave one thing out (timing-valid, numbers wrong)
mkdir -p gpurun_out
OUT=gpurun_out/r06_c0_ladder.txt
: > $OUT
export CPC_CONV0_BWD=3
for v in base ${VARIANTS:-1 2 3 4 8 12 15} base; do
  if [ $v = base ]; then unset CPC2_HIP_LIB; else export CPC2_HIP_LIB=$PWD/probes/c0_abl$v.so; fi
  timeout -k 10 200 python3 bench.py --config small --cpu-seconds 0 --also= --steps 12 --warmup 4 > gpurun_out/c0l_$v.json 2>gpurun_out/c0l_$v.err || { echo "$v FAILED" >> $OUT; tail -3 gpurun_out/c0l_$v.err >> $OUT; continue; }
  python3 - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/c0l_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("%-8s step %.3f ms  conv0_bwd %.1f us conv0_fwd %.1f us" % ("$v", d["ms_per_step"], 1e3*k["conv0_bwd"]["ms_per_step"], 1e3*k["conv0_fwd"]["ms_per_step"]))
PY
done
cat $OUT
