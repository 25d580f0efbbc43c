#!/bin/bash
# tools/ab_configs.sh OUTDIR REPS "A_ENV" "B_ENV" cfg1 cfg2 ...: alternating A/B of bench.py per configuration on ONE box
out=$1; reps=$2; a=$3; b=$4; shift 4
for cfg in "$@"; do
  bash tools/ab_env.sh $out/$cfg $reps "$a" "$b" --config $cfg --steps 16 --warmup 6 --cpu-seconds 0 --also= | sed "s/^/$cfg /"
done
