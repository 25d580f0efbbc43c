#!/bin/bash
OUT=gpurun_out/ab_pair_step.txt
: > $OUT
for v in pair nopair; do
  if [ $v = nopair ]; then export CPC_PLANES_NO_PAIR=1; else unset CPC_PLANES_NO_PAIR; fi
  for shape in fwd1 dgrad1; do
    case $shape in
      fwd1) export PROBE_TAPS=8 PROBE_STRIDE=4 PROBE_COLS=256 PROBE_L=1024;;
      dgrad1) export PROBE_TAPS=2 PROBE_STRIDE=1 PROBE_COLS=1024 PROBE_L=1024;;
    esac
    for d in 8 9; do
      echo "== $v $shape dbg $d" >> $OUT
      CPC_PLANES_DBG=$d PROBE_TN=0 timeout -k 10 120 python tools/planes_probe.py 3 2>&1 | grep -E "stamps" | tail -1 >> $OUT
    done
  done
done
unset PROBE_TAPS PROBE_STRIDE PROBE_COLS PROBE_L
for rep in 1 2 3; do
  for v in pair nopair; do
    if [ $v = nopair ]; then export CPC_PLANES_NO_PAIR=1; else unset CPC_PLANES_NO_PAIR; fi
    timeout -k 10 200 python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/ab_ps_$v.json 2>gpurun_out/ab_ps_$v.err || tail -5 gpurun_out/ab_ps_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_ps_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("step $v rep $rep: %.3f ms/step  planes_nt %.3f  planes_tn %.3f  frac %.3f" % (d["ms_per_step"], k["gemm_planes_nt"]["ms_per_step"], k["gemm_planes_tn"]["ms_per_step"], d["roofline"]["frac"]))
PY
  done
done
cat $OUT
