set -e
python -m pytest tests -m gpu -x -q -k "encoder or config_c or train_step or reproducible or reference_loss" > gpurun_out/nf_tests.log 2>&1 || { tail -30 gpurun_out/nf_tests.log | cut -c1-220; exit 1; }
tail -2 gpurun_out/nf_tests.log
for v in fused plain fused plain; do
  if [ $v = plain ]; then export CPC_NO_NORM_FUSION=1; else unset CPC_NO_NORM_FUSION; fi
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/nf_$v.json 2>gpurun_out/nf_$v.err || tail -5 gpurun_out/nf_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/nf_$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$v", d["ms_per_step"], d["value"], d["roofline"]["frac"], {n:v["ms_per_step"] for n,v in k.items() if "planes_nt" in n})
PY
done
