# one PMC pass (FETCH_SIZE) of the default bench:  tools/fetch_pass.sh <tag>
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_$1
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-seconds 0 --no-prof > $OUT/run.log 2>&1 || exit 1
