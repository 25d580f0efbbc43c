set -e
mkdir -p gpurun_out/ab
for cfg in large small; do
  export CPC2_HIP_LIB=$PWD/tools/variant/lib_grumm.so
  python bench.py --config $cfg --cpu-seconds 0 --no-prof --also "" --steps 8 --warmup 4 > gpurun_out/ab/mm_$cfg.json 2>gpurun_out/ab/mm_$cfg.err || true
  echo "== $cfg stamps"; grep "gru stamps" gpurun_out/ab/mm_$cfg.err | tail -8 | cut -c1-300
  unset CPC2_HIP_LIB
  python bench.py --config $cfg --cpu-seconds 0 --also "" --steps 20 --warmup 6 > gpurun_out/ab/mmmain_$cfg.json 2>gpurun_out/ab/mmmain_$cfg.err || tail -5 gpurun_out/ab/mmmain_$cfg.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/mmmain_$cfg.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$cfg main", d["ms_per_step"], d["value"], {n:(v["ms_per_step"], v["launches_per_step"]) for n,v in k.items() if "gru" in n})
PY
done
