CPC_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m cProfile -o gpurun_out/dp_prof.out bench.py --gpus 1 --steps 40 --warmup 5 --cpu-seconds 0 --also "" --no-prof > gpurun_out/dp_prof.json 2> gpurun_out/dp_prof.err
python - <<'PY'
import pstats
p=pstats.Stats("gpurun_out/dp_prof.out")
p.sort_stats("tottime").print_stats(22)
PY
