#!/bin/bash
# Process-group mode against the plain step on ONE box: un-profiled pairs (free and pinned to 2 cores), then one kernel trace of each
# mode (queue ids per dispatch).  bench.py is itself the rank (RANK / WORLD_SIZE set here): no spawn hop under the profiler.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r06_pg
mkdir -p $out
common="--steps 16 --warmup 6 --cpu-seconds 0 --also= "
dist_env="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 CPC_BENCH_FORCE_DIST=1"
for rep in 1 2; do
  python3 bench.py $common > $out/plain_$rep.json 2> $out/plain_$rep.err || exit 1
  env $dist_env MASTER_PORT=2950$rep python3 bench.py $common > $out/dist_$rep.json 2> $out/dist_$rep.err || exit 1
done
CPC_BENCH_PIN_CORES=2 python3 bench.py $common > $out/plain_pin2.json 2> $out/plain_pin2.err || exit 1
env $dist_env MASTER_PORT=29507 CPC_BENCH_PIN_CORES=2 python3 bench.py $common > $out/dist_pin2.json 2> $out/dist_pin2.err || exit 1
CPC_BENCH_PIN_CORES=1 python3 bench.py $common > $out/plain_pin1.json 2> $out/plain_pin1.err || exit 1
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 CPC_BENCH_FORCE_DIST=1
rocprofv3 --kernel-trace -d $out/kt_dist -o dist -- python3 bench.py --steps 10 --warmup 4 --cpu-seconds 0 --also= --no-prof > $out/dist_traced.json 2> $out/dist_traced.err || exit 1
unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT CPC_BENCH_FORCE_DIST
rocprofv3 --kernel-trace -d $out/kt_plain -o plain -- python3 bench.py --steps 10 --warmup 4 --cpu-seconds 0 --also= --no-prof > $out/plain_traced.json 2> $out/plain_traced.err || exit 1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_pg/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], d["ms_per_step"], json.dumps(d.get("host")), d["comm"].get("exposed_ms_per_step"))
PY
