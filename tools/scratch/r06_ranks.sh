mkdir -p gpurun_out
export CPC_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 4 --steps 4 --warmup 2 > gpurun_out/r06_bench_gloo4.json 2> gpurun_out/r06_bench_gloo4.err || { echo "4 ranks failed"; tail -5 gpurun_out/r06_bench_gloo4.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_bench_gloo4.json") if l.startswith("{")][-1])
print("4 gloo ranks on one GPU:", d["n_gpus"], d["ms_per_step"], d["value"], d["comm"]["world"], [ (o["config"]["workload"][:9], o.get("ms_per_step")) for o in d["other_configs"]], d["host"].get("rccl_stream_runs_beside_training_stream"))
PY
timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 2 --also "" > gpurun_out/r06_bench_spawn2.json 2> gpurun_out/r06_bench_spawn2.err || { echo "self-spawned 2 ranks failed"; tail -5 gpurun_out/r06_bench_spawn2.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_bench_spawn2.json") if l.startswith("{")][-1])
print("bench.py --gpus 2 starting its own ranks:", d["n_gpus"], d["ms_per_step"], d["value"])
PY
