run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 24 --warmup 8 --cpu-seconds 0 --also "" --no-prof > gpurun_out/dp4_$tag.json 2> gpurun_out/dp4_$tag.err; python - <<PY
import json
d=json.loads(open("gpurun_out/dp4_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"])
PY
}
run dist CPC_BENCH_FORCE_DIST=1
run dist_prio_default CPC_BENCH_FORCE_DIST=1 CPC_SIDE_PRIO_DEFAULT=1
run dist_nodefer_prio_default CPC_BENCH_FORCE_DIST=1 CPC_SIDE_PRIO_DEFAULT=1 CPC_NCE_NO_DEFER=1
run plain_prio_default CPC_SIDE_PRIO_DEFAULT=1
