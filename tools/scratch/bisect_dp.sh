ROOT=$PWD
for sha in 7332a76 88655af 4da758e 7cdaba4 HEAD; do
  if [ $sha = HEAD ]; then d=$ROOT; else d=$ROOT/tools/scratch/wt_$sha; fi
  cd $d
  for mode in plain dist; do
    unset CPC_BENCH_FORCE_DIST; [ $mode = dist ] && export CPC_BENCH_FORCE_DIST=1
    timeout -k 10 300 python bench.py --gpus 1 --steps 24 --warmup 8 --cpu-seconds 0 --also "" --no-prof > $ROOT/gpurun_out/bis_${sha}_$mode.json 2> $ROOT/gpurun_out/bis_${sha}_$mode.err
    python - <<PY
import json
try:
    d=json.loads(open("$ROOT/gpurun_out/bis_${sha}_$mode.json").read().strip().splitlines()[-1])
    print("$sha $mode", d["ms_per_step"])
except Exception as e:
    print("$sha $mode failed", e)
PY
  done
done
