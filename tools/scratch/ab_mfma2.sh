export CPC2_HIP_LIB=$PWD/tools/variant/lib_grumm.so
export CPC_NCE_NO_DEFER=1
python bench.py --config large --cpu-seconds 0 --no-prof --also "" --steps 8 --warmup 4 > gpurun_out/mm_large.json 2>gpurun_out/mm_large.err || true
grep "gru stamps" gpurun_out/mm_large.err | tail -4 | cut -c1-300
