set -e
python -m pytest tests -m gpu -x -q -k "gru or train_step or config_c or keep or reference_loss" > gpurun_out/fast_tests.log 2>&1 || { tail -25 gpurun_out/fast_tests.log | cut -c1-200; exit 1; }
tail -2 gpurun_out/fast_tests.log
for cfg in small large; do
  python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/fast_$cfg.json 2>gpurun_out/fast_$cfg.err || tail -5 gpurun_out/fast_$cfg.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/fast_$cfg.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$cfg", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n})
PY
done
