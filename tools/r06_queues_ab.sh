#!/bin/bash
# One-rank process-group step with the runtime's four hardware queues per priority against eight, alternating, 6 runs each.
out=gpurun_out/r06_q48
mkdir -p $out
common="--steps 16 --warmup 6 --cpu-seconds 0 --also="
for rep in 1 2 3 4 5 6; do
  for q in 4 8; do
    env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=297$q$rep CPC_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=$q python3 bench.py $common > $out/q${q}_$rep.json 2> $out/q${q}_$rep.err || exit 1
    python3 - $out/q${q}_$rep.json "queues $q rep $rep" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); h=d["host"]; k=d["kernels"]
print(sys.argv[2], d["ms_per_step"], "held_by_side", h.get("training_stream_held_by_side_stream_ms_per_step"), "gemm_nt", k["gemm_nt"]["ms_per_step"], "planes_nt", k["gemm_planes_nt"]["ms_per_step"], "gru_fwd", k["gru_fwd"]["ms_per_step"], flush=True)
PY
  done
done
