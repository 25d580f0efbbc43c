#!/bin/bash
# N one-rank process-group runs of the headline step on one box, one line each: ms per step, median step, the side-stream hold time,
# whether RCCL's stream runs beside the training stream, the box snapshot, class timers.
n=${1:-10}; q=${2:-8}
out=gpurun_out/r06_dist_runs
mkdir -p $out
for rep in $(seq 1 $n); do
  env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29800 + rep)) CPC_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 16 --warmup 6 --cpu-seconds 0 --also= > $out/run_$rep.json 2> $out/run_$rep.err || { echo "run $rep failed"; tail -3 $out/run_$rep.err; exit 1; }
  python3 - $out/run_$rep.json $rep <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); h=d["host"]; k=d["kernels"]
print("run", sys.argv[2], d["ms_per_step"], "median", h["step_ms_median"], "held", h.get("training_stream_held_by_side_stream_ms_per_step"), "rccl_beside", h.get("rccl_stream_runs_beside_training_stream"), "box", {x: h["gpu_box"].get(x) for x in ("sclk_level_mhz","mclk_level_mhz","fclk_level_mhz","power_w","other_gpus_busy")} if h.get("gpu_box") else None, "gru_fwd", k["gru_fwd"]["ms_per_step"], "gemm_nt", k["gemm_nt"]["ms_per_step"], flush=True)
PY
done
