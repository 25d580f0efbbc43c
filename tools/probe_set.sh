set -e
timeout -k 10 200 python tools/gemm_accuracy.py
timeout -k 10 100 python tools/gemm_probe.py tn 30
PROBE_M=98304 PROBE_TAPS=4 PROBE_STRIDE=2 timeout -k 10 100 python tools/gemm_probe.py tn 30
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -5
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null
