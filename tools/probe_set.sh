set -e
P="timeout -k 10 100 python tools/gemm_probe.py nt 20"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
PROBE_M=131072 PROBE_TAPS=8 PROBE_STRIDE=4 $P
PROBE_M=131072 PROBE_TAPS=4 PROBE_STRIDE=2 $P
PROBE_M=131072 PROBE_TAPS=2 PROBE_STRIDE=1 $P
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null
