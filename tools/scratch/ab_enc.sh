set -e
mkdir -p gpurun_out/ab
for v in two one two one; do
  if [ $v = one ]; then export CPC_ENC_ONE_STREAM=1; else unset CPC_ENC_ONE_STREAM; fi
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/ab/enc_$v.json 2>gpurun_out/ab/enc_$v.err || tail -5 gpurun_out/ab/enc_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/enc_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["value"], d["roofline"]["frac"])
PY
done
unset CPC_ENC_ONE_STREAM
python -m pytest tests -m gpu -x -q -k "train_step or config or dp or reproducible or encoder" > gpurun_out/ab/enc_tests.log 2>&1; tail -3 gpurun_out/ab/enc_tests.log
bash tools/trace_step.sh r03l --also "" && python tools/show_trace.py gpurun_out/kt_r03l > gpurun_out/kt_r03l.txt
