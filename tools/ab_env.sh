#!/bin/bash
# Alternating A/B of bench.py under two environments on ONE box:  tools/ab_env.sh OUTDIR REPS "A_ENV=1 ..." "B_ENV=1 ..." [bench args]
# An empty string is the plain environment.  Prints ms/step per run.
out=$1; reps=$2; a=$3; b=$4; shift 4
mkdir -p $out
args=${@:---steps 16 --warmup 6 --cpu-seconds 0 --also=}
for rep in $(seq 1 $reps); do
  for arm in A B; do
    envs=$a; [ $arm = B ] && envs=$b
    env $envs python3 bench.py $args > $out/${arm}_$rep.json 2> $out/${arm}_$rep.err || { echo "$arm $rep failed"; tail -5 $out/${arm}_$rep.err; exit 1; }
    python3 - $out/${arm}_$rep.json "$arm[$envs] rep $rep" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels",{})
print(sys.argv[2], d["ms_per_step"], "ms/step", " ".join(f"{n}={v['ms_per_step']}" for n,v in k.items()))
PY
  done
done
