for v in behind beside nodefer behind beside nodefer; do
  unset CPC_NCE_NO_DEFER CPC_NCE_BESIDE_GRU
  [ $v = nodefer ] && export CPC_NCE_NO_DEFER=1
  [ $v = beside ] && export CPC_NCE_BESIDE_GRU=1
  python bench.py --cpu-seconds 0 --also "" --steps 40 --warmup 10 > gpurun_out/df2_$v.json 2>gpurun_out/df2_$v.err || tail -5 gpurun_out/df2_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/df2_$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$v", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru_bwd" in n or n in ("gemm_tn","gemm_nt")})
PY
done
