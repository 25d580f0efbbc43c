mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gru or lstm or rnn or recurrent or CPCAR or ar_ or train_step or config" > gpurun_out/tr_tests.log 2>&1 || { tail -30 gpurun_out/tr_tests.log; exit 1; }
tail -2 gpurun_out/tr_tests.log
for cfg in large small recipe; do for rep in 1 2; do
  python3 bench.py --config $cfg --steps 16 --warmup 6 --cpu-seconds 0 --also= > gpurun_out/tr_$cfg$rep.json 2> gpurun_out/tr_$cfg$rep.err || { tail -5 gpurun_out/tr_$cfg$rep.err; exit 1; }
  python3 - gpurun_out/tr_$cfg$rep.json $cfg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["host"]["step_ms_median"])
PY
done; done
