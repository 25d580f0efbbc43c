set -e
P="timeout -k 10 100 python tools/gemm_probe.py nt 10"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm" 2>&1 | tail -3
PROBE_M=131072 PROBE_TAPS=8 PROBE_STRIDE=4 $P
PROBE_M=131072 PROBE_TAPS=4 PROBE_STRIDE=2 $P
CPC_X6_PIPE_MINK=512 PROBE_M=131072 PROBE_TAPS=2 PROBE_STRIDE=1 $P
PROBE_M=131072 PROBE_TAPS=32 PROBE_STRIDE=4 $P
