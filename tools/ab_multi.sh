#!/bin/bash
# tools/ab_multi.sh OUTDIR REPS "ENV_A" "ENV_B" "ENV_C" ... -- bench args...   : alternating arms of bench.py on ONE box (class timers printed)
out=$1; reps=$2; shift 2
arms=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do arms+=("$1"); shift; done
shift
args=${@:---steps 16 --warmup 6 --cpu-seconds 0 --also=}
mkdir -p $out
for rep in $(seq 1 $reps); do
  i=0
  for envs in "${arms[@]}"; do
    env $envs python3 bench.py $args > $out/arm${i}_$rep.json 2> $out/arm${i}_$rep.err || { echo "arm $i rep $rep failed"; tail -5 $out/arm${i}_$rep.err; exit 1; }
    python3 - $out/arm${i}_$rep.json "[$envs] rep $rep" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels",{})
print(sys.argv[2], d["ms_per_step"], "ms/step median", d["host"]["step_ms_median"], " ".join(f"{n}={v['ms_per_step']}" for n,v in k.items()), flush=True)
PY
    i=$((i+1))
  done
done
