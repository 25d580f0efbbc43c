# per-dispatch timeline of the last bench step: bash tools/trace_step.sh <tag> [bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-seconds 0 --no-prof "$@" > $OUT.log 2>&1
ls $OUT/*/
