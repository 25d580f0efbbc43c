#!/bin/bash
# kernel statistics + one step's timeline per BASELINE configuration on the round's last build
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in small large transformer; do
  OUT=gpurun_out/prof_r06c_$cfg
  mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --config $cfg --steps 3 --warmup 2 --cpu-seconds 0 --no-prof --also= > $OUT/stats.log 2>&1 || { echo "stats $cfg failed"; tail -5 $OUT/stats.log; exit 1; }
  f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/r06_c_${cfg}_kernel_stats.csv
  bash tools/trace_step.sh r06c_$cfg --config $cfg --also= > /dev/null 2>&1 || { echo "trace $cfg failed"; exit 1; }
  python3 tools/show_trace.py gpurun_out/kt_r06c_$cfg > gpurun_out/r06_c_${cfg}_step_timeline.txt 2>&1
  grep "kernel sum\|launches" gpurun_out/r06_c_${cfg}_step_timeline.txt | head -2
done
head -5 gpurun_out/r06_c_small_kernel_stats.csv | cut -c1-200
