# kernel-level A/B of the criterion's dz path
tools/profile_config.sh small || exit 1
mv gpurun_out/prof_cfg_small gpurun_out/prof_cfg_small_v
CPC_NCE_ATOMIC=1 tools/profile_config.sh small || exit 1
mv gpurun_out/prof_cfg_small gpurun_out/prof_cfg_small_a
