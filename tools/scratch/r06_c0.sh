mkdir -p gpurun_out
for f in 0 2 3; do
  CPC_CONV0_BWD=$f timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "encoder" > gpurun_out/c0_tests_$f.log 2>&1 || { echo "tests form $f failed"; tail -30 gpurun_out/c0_tests_$f.log; exit 1; }
  tail -1 gpurun_out/c0_tests_$f.log
done
for rep in 1 2; do for f in 0 2 3; do
  CPC_CONV0_BWD=$f python3 bench.py --steps 16 --warmup 6 --cpu-seconds 0 --also= > gpurun_out/c0_bench_${f}_$rep.json 2> gpurun_out/c0_bench_${f}_$rep.err || { echo "bench $f failed"; tail -5 gpurun_out/c0_bench_${f}_$rep.err; exit 1; }
  python3 - gpurun_out/c0_bench_${f}_$rep.json "form $f rep $rep" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels",{})
print(sys.argv[2], d["ms_per_step"], "ms/step", " ".join(f"{n}={v['ms_per_step']}" for n,v in k.items() if 'conv0' in n), d.get("final_loss"))
PY
done; done
