set -e
P="timeout -k 10 100 python tools/gemm_probe.py nt 20"
PROBE_M=98304 PROBE_TAPS=8 PROBE_STRIDE=4 $P
PROBE_M=98304 PROBE_TAPS=2 PROBE_STRIDE=1 $P
