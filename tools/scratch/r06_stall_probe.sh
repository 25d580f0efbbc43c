#!/bin/bash
# the rare single long step (70-165 ms, ~1 per 20 000 steps): the garbage collector?  40 000 steps with and without it
mkdir -p gpurun_out
export CPC_BENCH_FORCE_DIST=1
for gc in on off; do
  if [ $gc = off ]; then PRE="import gc; gc.disable(); "; else PRE=""; fi
  timeout -k 10 400 python -c "${PRE}import runpy; runpy.run_path('bench.py', run_name='__main__')" --config small --steps 40000 --warmup 5 --no-prof --cpu-seconds 0 --also "" > gpurun_out/stall_$gc.json 2> gpurun_out/stall_$gc.err || { echo "$gc failed"; tail -3 gpurun_out/stall_$gc.err; continue; }
  python - $gc <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/stall_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host"]
print("gc", sys.argv[1], d["ms_per_step"], "median", h["step_ms_median"], "max", h["step_ms_max"], "at", h["step_ms_max_index"], "over 2x:", h["steps_over_2x_median"])
PY
done
