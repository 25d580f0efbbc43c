for v in slab noslab slab noslab; do
  if [ $v = noslab ]; then export CPC_NO_SLAB_NORM=1; else unset CPC_NO_SLAB_NORM; fi
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/sl_$v.json 2>gpurun_out/sl_$v.err || tail -5 gpurun_out/sl_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sl_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["value"])
PY
done
