#!/bin/bash
# the persistent pair kernel (next item's first double stage requested under the epilogue) against one workgroup per tile
OUT=gpurun_out/ab_persist.txt
: > $OUT
for rep in 1 2; do
  for v in persist tile; do
    if [ $v = tile ]; then export CPC_PLANES_NO_PERSIST=1; else unset CPC_PLANES_NO_PERSIST; fi
    for shape in dgrad1 dgrad2 fwd3; do
      case $shape in
        dgrad1) export PROBE_TAPS=2 PROBE_STRIDE=1 PROBE_COLS=1024 PROBE_L=1024;;
        dgrad2) export PROBE_TAPS=2 PROBE_STRIDE=1 PROBE_COLS=512 PROBE_L=512;;
        fwd3) export PROBE_TAPS=4 PROBE_STRIDE=2 PROBE_COLS=256 PROBE_L=256;;
      esac
      echo "== $v rep $rep $shape" >> $OUT
      PROBE_TN=0 timeout -k 10 120 python tools/planes_probe.py 20 2>&1 | grep -E "planes nt|fp64" >> $OUT
    done
  done
done
unset PROBE_TAPS PROBE_STRIDE PROBE_COLS PROBE_L
for rep in 1 2 3; do
  for v in persist tile; do
    if [ $v = tile ]; then export CPC_PLANES_NO_PERSIST=1; else unset CPC_PLANES_NO_PERSIST; fi
    timeout -k 10 200 python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/ab_pp_$v.json 2>gpurun_out/ab_pp_$v.err || tail -5 gpurun_out/ab_pp_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_pp_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("step $v rep $rep: %.3f ms/step  planes_nt %.3f  frac %.3f" % (d["ms_per_step"], k["gemm_planes_nt"]["ms_per_step"], d["roofline"]["frac"]))
PY
  done
done
cat $OUT
