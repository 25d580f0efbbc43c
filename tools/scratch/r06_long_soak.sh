#!/bin/bash
# one long run per mode on the round's last build: rare events (a cooperative kernel's timeout, a sampler race, a queue collision) need many steps
mkdir -p gpurun_out
OUT=${OUT:-gpurun_out/r06_long_soak.txt}
: > $OUT
IFS=";" read -ra SPECLIST <<< "${SPECS:-small 50000 1;small 20000 0;large 6000 1}"
for spec in "${SPECLIST[@]}"; do
  set -- $spec
  if [ $3 = 1 ]; then export CPC_BENCH_FORCE_DIST=1; else unset CPC_BENCH_FORCE_DIST; fi
  timeout -k 10 420 python bench.py --config $1 --steps $2 --warmup 5 --no-prof --cpu-seconds 0 --also "" > gpurun_out/lsoak.json 2> gpurun_out/lsoak.err || { echo "$spec FAILED" >> $OUT; tail -3 gpurun_out/lsoak.err >> $OUT; continue; }
  python - "$1" "$2" "$3" >> $OUT <<'PY'
import json, sys
d = json.loads(open("gpurun_out/lsoak.json").read().strip().splitlines()[-1])
h = d["host"]
print("%-6s %6s steps  process group %s  %8.3f ms/step  median %.3f  max %.3f (step %d)  over 2x median: %d   final losses %s" % (
    sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"], h["step_ms_median"], h["step_ms_max"], h["step_ms_max_index"],
    len(h["steps_over_2x_median"]), " ".join("%.4f" % v for v in d["config"]["final_losses"][:3])))
PY
  tail -1 $OUT
done
