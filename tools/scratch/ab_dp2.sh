for v in plain dist plain dist; do
  unset CPC_BENCH_FORCE_DIST
  if [ $v = dist ]; then export CPC_BENCH_FORCE_DIST=1; fi
  timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 10 --cpu-seconds 0 --also "" > gpurun_out/dp2_$v.json 2> gpurun_out/dp2_$v.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("gpurun_out/dp2_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$v", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n or "planes" in n or "conv0" in n})
PY
done
