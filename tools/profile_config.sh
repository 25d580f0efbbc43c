#!/usr/bin/env bash
# kernel-trace summary of one bench configuration:  tools/profile_config.sh <config> [extra bench args]
set -u
CFG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $CFG -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 3 --warmup 2 --cpu-seconds 0 --no-prof "$@" > $OUT/run.log 2>&1 || exit 1
find $OUT -name "*kernel_stats.csv" | head -3
