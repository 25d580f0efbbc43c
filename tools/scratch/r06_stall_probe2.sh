#!/bin/bash
# the same 40 000 steps with the collector held as bench.py / train.run() now do (gc.freeze after the warm-up), and once more untouched
mkdir -p gpurun_out
export CPC_BENCH_FORCE_DIST=1
for mode in frozen untouched; do
  if [ $mode = untouched ]; then export CPC_BENCH_NO_GC_FREEZE=1; else unset CPC_BENCH_NO_GC_FREEZE; fi
  timeout -k 10 400 python bench.py --config small --steps 40000 --warmup 5 --no-prof --cpu-seconds 0 --also "" > gpurun_out/stall_$mode.json 2> gpurun_out/stall_$mode.err || { echo "$mode failed"; tail -3 gpurun_out/stall_$mode.err; continue; }
  python - $mode <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/stall_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host"]
print("collector", sys.argv[1], h.get("python_gc"), d["ms_per_step"], "median", h["step_ms_median"], "max", h["step_ms_max"], "at", h["step_ms_max_index"], "over 2x:", h["steps_over_2x_median"])
PY
done
