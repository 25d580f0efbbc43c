#!/bin/bash
# Similarity kernel (infonce_fwd_dma_kernel<256>) inside the CPC-small step: probe builds that leave one thing out
# (timing-valid, numbers wrong): probes/nce_*.so built by tools/build_variant.sh infonce.hip -DNCE_DBG=.. -DNCE_ABL=..
OUT=gpurun_out/r05_nce_ladder.txt
: > $OUT
for v in base abl4 hot16 l2 l2_abl4 hot16_abl4 nop l2_nop l2_nop_nostore nostore base; do
  if [ $v = base ]; then unset CPC2_HIP_LIB; else export CPC2_HIP_LIB=$PWD/probes/nce_$v.so; fi
  timeout -k 10 200 python bench.py --config small --cpu-seconds 0 --also "" --steps 12 --warmup 4 > gpurun_out/nce_$v.json 2>gpurun_out/nce_$v.err || { echo "$v FAILED" >> $OUT; tail -3 gpurun_out/nce_$v.err >> $OUT; continue; }
  python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/nce_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("%-16s step %.3f ms  infonce_fwd %.1f us  infonce_bwd %.1f us" % ("$v", d["ms_per_step"], 1e3*k["infonce_fwd"]["ms_per_step"], 1e3*k["infonce_bwd"]["ms_per_step"]))
PY
done
cat $OUT
