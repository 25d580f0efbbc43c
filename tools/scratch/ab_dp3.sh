run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 10 --cpu-seconds 0 --also "" > gpurun_out/dp3_$tag.json 2> gpurun_out/dp3_$tag.err; python - <<PY
import json
d=json.loads(open("gpurun_out/dp3_$tag.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$tag", d["ms_per_step"], {n:v["ms_per_step"] for n,v in k.items() if "planes_nt" in n or "conv0_fwd" in n})
PY
}
run plain X=1
run nccl1 CPC_BENCH_FORCE_DIST=1
run gloo1 CPC_BENCH_FORCE_DIST=1 CPC_BENCH_BACKEND=gloo
run nccl1_noprof CPC_BENCH_FORCE_DIST=1 CPC_X=1
