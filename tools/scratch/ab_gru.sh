set -e
mkdir -p gpurun_out/ab
for cfg in small large; do
for v in main gruabl; do
  if [ $v = main ]; then unset CPC2_HIP_LIB; else export CPC2_HIP_LIB=$PWD/tools/variant/lib_$v.so; fi
  python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 8 --warmup 4 > gpurun_out/ab/gru_$cfg$v.json 2>gpurun_out/ab/gru_$cfg$v.err || { tail -5 gpurun_out/ab/gru_$cfg$v.err; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/gru_$cfg$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$cfg $v", d["ms_per_step"], {n:(v["ms_per_step"], v["launches_per_step"]) for n,v in k.items() if "gru" in n})
PY
done; done
