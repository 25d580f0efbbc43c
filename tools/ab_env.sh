# A/B of one environment switch on the default bench:  tools/ab_env.sh VAR [bench args]
VAR=$1; shift
for i in 1 2; do
  python bench.py --cpu-seconds 0 --steps 40 --warmup 8 "$@" > gpurun_out/ab_on.json 2>/dev/null && python tools/bench_kernels.py gpurun_out/ab_on.json gemm
  env $VAR=1 python bench.py --cpu-seconds 0 --steps 40 --warmup 8 "$@" > gpurun_out/ab_off.json 2>/dev/null && echo "  with $VAR=1:" && python tools/bench_kernels.py gpurun_out/ab_off.json gemm
done
