#!/usr/bin/env bash
# rocprofv3 evidence run: kernel stats + three PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy), each its own run of the same
# bench.py command.   tools/gpu_profile.sh <tag> [config]     env BENCH_ARGS: extra bench.py arguments
set -u
OUT=gpurun_out/prof_$1
CFG=${2:-small}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --config $CFG --steps 3 --warmup 2 --cpu-seconds 0 --no-prof --also= ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma -- $CMD > $OUT/mfma.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
