export CPC_NCE_NO_DEFER=1
for v in mfma nomfma; do
  if [ $v = nomfma ]; then export CPC_GRU_NO_MFMA=1; else unset CPC_GRU_NO_MFMA; fi
  python bench.py --config large --cpu-seconds 0 --also "" --steps 12 --warmup 4 > gpurun_out/mf3_$v.json 2>gpurun_out/mf3_$v.err || tail -5 gpurun_out/mf3_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/mf3_$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("no-defer $v", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n})
PY
done
unset CPC_NCE_NO_DEFER CPC_GRU_NO_MFMA
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_c.log 2>&1; tail -3 gpurun_out/r03_gpu_tests_c.log
