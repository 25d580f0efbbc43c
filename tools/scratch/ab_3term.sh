set -e
python -m pytest tests -m gpu -x -q -s -k "three_term or reference_loss_curve or gemm" > gpurun_out/t3_tests.log 2>&1 || { tail -25 gpurun_out/t3_tests.log | cut -c1-200; exit 1; }
grep -h "three-term mode vs\|passed" gpurun_out/t3_tests.log | tail -3
for cfg in small small_3term; do
  python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/t3_$cfg.json 2>gpurun_out/t3_$cfg.err || tail -5 gpurun_out/t3_$cfg.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/t3_$cfg.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$cfg", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["peak"], {n:v["ms_per_step"] for n,v in k.items() if "planes" in n}, d["config"]["final_losses"][:3])
PY
done
