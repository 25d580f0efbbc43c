set -e
mkdir -p gpurun_out/ab
for v in head a1 main a2 a4 head main; do
  if [ $v = main ]; then unset CPC2_HIP_LIB; else export CPC2_HIP_LIB=$PWD/tools/variant/lib_$v.so; fi
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/ab/$v.json 2>gpurun_out/ab/$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or d.get("kernel_ms") or {}
print("$v", d["ms_per_step"], {n:v for n,v in (k.items() if isinstance(k,dict) else []) if "conv0" in n or "enc" in n})
PY
done
CPC2_HIP_LIB= python -m pytest tests -m gpu -x -q -k "encoder or config_c or train_step" > gpurun_out/ab/tests.log 2>&1; tail -3 gpurun_out/ab/tests.log
