set -e
P="timeout -k 10 100 python tools/gemm_probe.py nt 20"
cp cpc2_amd/libcpc2_hip.so /tmp/base.so
for v in 1 2; do
cp tools/abl/lib$v.so cpc2_amd/libcpc2_hip.so
echo "variant $v"
PROBE_M=98304 PROBE_TAPS=8 PROBE_STRIDE=4 $P
done
cp /tmp/base.so cpc2_amd/libcpc2_hip.so
