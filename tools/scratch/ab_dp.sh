for v in defer nodefer defer_nooverlap; do
  unset CPC_NCE_NO_DEFER CPC_BENCH_NO_OVERLAP
  if [ $v = nodefer ]; then export CPC_NCE_NO_DEFER=1; fi
  if [ $v = defer_nooverlap ]; then export CPC_BENCH_NO_OVERLAP=1; fi
  CPC_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 10 --cpu-seconds 0 --also "" > gpurun_out/dp_$v.json 2> gpurun_out/dp_$v.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("gpurun_out/dp_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$v", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n or "planes" in n})
PY
done
