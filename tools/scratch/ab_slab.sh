set -e
python -m pytest tests -m gpu -x -q -k "encoder or config_c or train_step or reproducible or reference_loss or three_term" > gpurun_out/sl_tests.log 2>&1 || { tail -30 gpurun_out/sl_tests.log | cut -c1-220; exit 1; }
tail -2 gpurun_out/sl_tests.log
for i in 1 2; do
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/sl.json 2>gpurun_out/sl.err || tail -5 gpurun_out/sl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sl.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"])
PY
done
