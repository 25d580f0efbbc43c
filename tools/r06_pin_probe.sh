#!/bin/bash
# One kernel trace of the one-rank process-group step pinned to two cores (queue ids per dispatch), then un-profiled pairs.
export TMPDIR=/tmp
out=gpurun_out/r06_pin
mkdir -p $out
common="--steps 16 --warmup 6 --cpu-seconds 0 --also="
for rep in 1 2 3; do
  env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2960$rep CPC_BENCH_FORCE_DIST=1 CPC_BENCH_PIN_CORES=2 python3 bench.py $common > $out/dist_pin2_$rep.json 2> $out/dist_pin2_$rep.err || exit 1
  env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2961$rep CPC_BENCH_FORCE_DIST=1 python3 bench.py $common > $out/dist_free_$rep.json 2> $out/dist_free_$rep.err || exit 1
done
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 CPC_BENCH_FORCE_DIST=1 CPC_BENCH_PIN_CORES=2
rocprofv3 --kernel-trace --output-format csv -d $out/kt_dist_pin2 -o t -- python3 bench.py --steps 10 --warmup 4 --cpu-seconds 0 --also= --no-prof > $out/dist_pin2_traced.json 2> $out/dist_pin2_traced.err || exit 1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_pin/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); h=d["host"]
    print(f.split("/")[-1], d["ms_per_step"], {k:h.get(k) for k in ("step_ms_median","step_ms_max","busy_ms_per_step","training_stream_held_by_side_stream_ms_per_step","side_stream_runs_beside_training_stream","sampler_stream_runs_beside_training_stream","exchange_helper_stream_runs_beside_training_stream","streams_handed_out_untested")}, {k:v["ms_per_step"] for k,v in d.get("kernels",{}).items()})
PY
python3 tools/show_queues.py $out/kt_dist_pin2 | head -12
