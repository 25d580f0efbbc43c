set -e
mkdir -p gpurun_out/ab
for v in defer nodefer defer nodefer; do
  if [ $v = nodefer ]; then export CPC_NCE_NO_DEFER=1; else unset CPC_NCE_NO_DEFER; fi
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/ab/df_$v.json 2>gpurun_out/ab/df_$v.err || tail -5 gpurun_out/ab/df_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/df_$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$v", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n or "infonce" in n})
PY
done
unset CPC_NCE_NO_DEFER
python -m pytest tests -m gpu -x -q -k "train_step or config or dp or reproducible or criterion or infonce" > gpurun_out/ab/df_tests.log 2>&1; tail -3 gpurun_out/ab/df_tests.log
