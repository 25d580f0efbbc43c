# HIP API calls between kernels that start after an idle gap on the training stream's queue: bash tools/trace_joins.sh <tag> [bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/hj_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-seconds 0 --no-prof "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_joins.py $OUT
