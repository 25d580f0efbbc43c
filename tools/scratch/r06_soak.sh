#!/bin/bash
# soak runs on the round's last build + shapes around the bench default + a leak check (tools/stress.sh)
OUT=gpurun_out/r06_soak.txt
echo "config        steps   ms/step   first three of the final losses" > $OUT
for spec in "small 2000" "large 500" "transformer 1000" "lstm 800" "small_strict 500" "recipe 300" "small_feeder 20"; do
  set -- $spec
  timeout -k 10 400 env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29741 CPC_BENCH_FORCE_DIST=1 python bench.py --config $1 --steps $2 --warmup 5 --no-prof --cpu-seconds 0 --also "" > gpurun_out/soak.json 2> gpurun_out/soak.err || { echo "$1 FAILED" >> $OUT; tail -3 gpurun_out/soak.err >> $OUT; continue; }
  python - "$1" "$2" >> $OUT <<'PY'
import json, sys
d = json.loads(open("gpurun_out/soak.json").read().strip().splitlines()[-1])
print("%-13s %5s  %8.3f    %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], " ".join("%.4f" % v for v in d["config"]["final_losses"][:3])))
PY
done
bash tools/stress.sh >> $OUT 2>&1
cat $OUT
