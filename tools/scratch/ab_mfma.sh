set -e
mkdir -p gpurun_out/ab
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "test_gru_vs_oracle_fp64" > gpurun_out/ab/mf_tests.log 2>&1 || { tail -25 gpurun_out/ab/mf_tests.log | cut -c1-200; exit 1; }
tail -2 gpurun_out/ab/mf_tests.log
for v in mfma nomfma mfma nomfma; do
  if [ $v = nomfma ]; then export CPC_GRU_NO_MFMA=1; else unset CPC_GRU_NO_MFMA; fi
  python bench.py --config large --cpu-seconds 0 --also "" --steps 12 --warmup 4 > gpurun_out/ab/mf_$v.json 2>gpurun_out/ab/mf_$v.err || tail -5 gpurun_out/ab/mf_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/mf_$v.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print("$v", d["ms_per_step"], d["value"], {n:v["ms_per_step"] for n,v in k.items() if "gru" in n})
PY
done
