for v in main c0w main c0w; do
  if [ $v = main ]; then unset CPC2_HIP_LIB; else export CPC2_HIP_LIB=$PWD/tools/variant/lib_$v.so; fi
  for cfg in small large; do
  python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 30 --warmup 8 > gpurun_out/c0w_$v$cfg.json 2>gpurun_out/c0w_$v$cfg.err || tail -5 gpurun_out/c0w_$v$cfg.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/c0w_$v$cfg.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$v $cfg", d["ms_per_step"], {n:v["ms_per_step"] for n,v in k.items() if "conv0" in n})
PY
  done
done
export CPC2_HIP_LIB=$PWD/tools/variant/lib_c0w.so
python -m pytest tests -m gpu -x -q -k "encoder or config_c or train_step" > gpurun_out/c0w_tests.log 2>&1; tail -2 gpurun_out/c0w_tests.log
