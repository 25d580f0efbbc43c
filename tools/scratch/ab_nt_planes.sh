#!/bin/bash
# plain NT products (context / predictor / transformer projections) through the plane-fed kernel (operands split by two streaming
# passes first: CPC_NT_AUTO_PLANES=<min GFLOP>) against the split-in-kernel family; alternating pairs on one box; first a parity check
OUT=gpurun_out/ab_nt_planes.txt
: > $OUT
CPC_NT_AUTO_PLANES=0 CPC_SKIP_DP_JOBS=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "transformer or train_step or gru or criterion or infonce" > gpurun_out/ab_nt_planes_tests.log 2>&1
echo "tests with CPC_NT_AUTO_PLANES=0: rc $? $(tail -1 gpurun_out/ab_nt_planes_tests.log)" >> $OUT
for cfg in small transformer large; do
for rep in 1 2; do
  for v in off on; do
    if [ $v = on ]; then export CPC_NT_AUTO_PLANES=${MINGF:-0}; else unset CPC_NT_AUTO_PLANES; fi
    timeout -k 10 200 python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/ab_n_$v.json 2>gpurun_out/ab_n_$v.err || tail -5 gpurun_out/ab_n_$v.err >> $OUT
    python - >> $OUT <<PY
import json
d=json.loads(open("gpurun_out/ab_n_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$cfg $v rep $rep: %.3f ms/step" % d["ms_per_step"], {n: (round(v["ms_per_step"], 3), v["launches_per_step"]) for n, v in k.items() if "gemm" in n})
PY
  done
done
done
unset CPC_NT_AUTO_PLANES
cat $OUT
