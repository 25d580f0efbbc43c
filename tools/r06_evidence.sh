#!/bin/bash
# Round-6 evidence on ONE box: kernel stats + PMC passes for the three BASELINE configurations, conv0's counters, one step timeline each.
set -u
for cfg in ${PROFILE_CFGS:-small large transformer}; do
  tag=r06a; [ $cfg != small ] && tag=r06a_$cfg
  bash tools/gpu_profile.sh $tag $cfg > gpurun_out/prof_$tag.log 2>&1 || { echo "profile $cfg failed"; tail -5 gpurun_out/prof_$tag.log; exit 1; }
  echo "profiled $cfg"
done
bash tools/pmc_kernels.sh r06a "conv0_fwd_pl_kernel|conv0_bwd_kernel|norm_bwd_pl_kernel|gemm_tn_x6p_kernel|gemm_nt_x6_kernel" small > gpurun_out/r06a_conv0_counters.txt 2>&1 || { echo "pmc kernels failed"; tail -5 gpurun_out/r06a_conv0_counters.txt; exit 1; }
for cfg in small large transformer; do
  bash tools/trace_step.sh r06a_$cfg --config $cfg --also= > /dev/null 2>&1 || { echo "trace $cfg failed"; exit 1; }
  python3 tools/show_trace.py gpurun_out/kt_r06a_$cfg > gpurun_out/r06a_${cfg}_step_timeline.txt 2>&1
done
CPC_PLANES_DBG=8 python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-prof --also= > /dev/null 2> gpurun_out/r06a_stamps.log
python3 tools/planes_shapes.py gpurun_out/kt_r06a_small gpurun_out/r06a_stamps.log > gpurun_out/r06a_planes_shapes.md 2>&1
tail -16 gpurun_out/r06a_planes_shapes.md
