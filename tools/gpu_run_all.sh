#!/usr/bin/env bash
# One GPU-box session: parity tests -> smoke -> short bench -> rocprof kernel stats.
# Stops at the first step that times out or dies from a signal (never starts another GPU step after that).
set -u
OUT=gpurun_out
mkdir -p $OUT
run() {   # run <name> <timeout_s> <cmd...>
    local name=$1 tmo=$2; shift 2
    echo "=== $name: $*" | tee -a $OUT/session.log
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2>&1
    local rc=$?
    echo "=== $name exit $rc" | tee -a $OUT/session.log
    tail -n 25 $OUT/$name.log
    if [ $rc -ge 124 ]; then echo "!!! $name timed out or was killed: stopping" | tee -a $OUT/session.log; exit $rc; fi
    return $rc
}
STEPS=${STEPS:-"tests smoke bench prof"}
for s in $STEPS; do
  case $s in
    tests) run pytest_gpu 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 ;;
    prof)  export TMPDIR=/tmp; run rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-prof
           find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats.csv ;;
  esac
done
exit 0
