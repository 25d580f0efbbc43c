for cap in 0 1024 512 256 0 512; do
  export CPC_NCE_GATHER_BLOCKS=$cap
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/cap_$cap.json 2>gpurun_out/cap_$cap.err || tail -5 gpurun_out/cap_$cap.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/cap_$cap.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("cap $cap", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n})
PY
done
