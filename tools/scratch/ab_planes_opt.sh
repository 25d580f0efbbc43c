#!/bin/bash
# A/B of the plane-fed kernels' schedule switches (CPC_PLANES_OPT bits) on the probe shapes and on the step
#   bash tools/scratch/ab_planes_opt.sh "0 1 2 3"
OPTS=${1:-"0 1"}
OUT=gpurun_out/ab_planes_opt.txt
: > $OUT
for rep in 1 2; do
  for o in $OPTS; do
    echo "== opt $o rep $rep: conv1 forward + weight gradient (random data)" >> $OUT
    CPC_PLANES_OPT=$o timeout -k 10 120 python tools/planes_probe.py 20 2>&1 | grep -E "planes (nt|tn)|fp64" >> $OUT
    echo "== opt $o rep $rep: conv1 backward data shape (K = 512, N = 1024)" >> $OUT
    CPC_PLANES_OPT=$o PROBE_TAPS=2 PROBE_STRIDE=1 PROBE_COLS=1024 PROBE_TN=0 timeout -k 10 120 python tools/planes_probe.py 20 2>&1 | grep -E "planes (nt|tn)|fp64" >> $OUT
  done
done
for rep in 1 2; do
  for o in $OPTS; do
    CPC_PLANES_OPT=$o timeout -k 10 200 python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/ab_po_$o.json 2>gpurun_out/ab_po_$o.err || tail -5 gpurun_out/ab_po_$o.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_po_$o.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("step opt $o rep $rep: %.3f ms/step  planes_nt %.3f  planes_tn %.3f  frac %.3f" % (d["ms_per_step"], k["gemm_planes_nt"]["ms_per_step"], k["gemm_planes_tn"]["ms_per_step"], d["roofline"]["frac"]))
PY
  done
done
cat $OUT
