set -e
CPC_GRU_MFMA_ALL=1 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "gru or keep or bidir or train_step" > gpurun_out/mf4_tests.log 2>&1 || { tail -25 gpurun_out/mf4_tests.log | cut -c1-200; exit 1; }
tail -2 gpurun_out/mf4_tests.log
export CPC_NCE_NO_DEFER=1
for v in all default all default; do
  if [ $v = all ]; then export CPC_GRU_MFMA_ALL=1; else unset CPC_GRU_MFMA_ALL; fi
  python bench.py --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/mf4_$v.json 2>gpurun_out/mf4_$v.err || tail -5 gpurun_out/mf4_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/mf4_$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("small no-defer $v", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n})
PY
done
