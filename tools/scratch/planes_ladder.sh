#!/bin/bash
# Upper bounds of the plane-fed kernel's levers (timing-valid, numbers-wrong probes; CPC_PLANES_DBG bits; a second run with + 8
# reads the in-kernel clock):  0 the kernel; 1 no requests after the prologue; 2 no MFMAs; 16 half the A requests (lever a: A staged
# once per tap pair); 32 a third fewer fragment reads (lever b: 128 x 128 wave tiles); 48 both
OUT=gpurun_out/planes_ladder.txt
: > $OUT
for rep in 1 2; do
  for d in 0 16 32 48 1 2; do
    for shape in fwd dgrad; do
      if [ $shape = fwd ]; then export PROBE_TAPS=8 PROBE_STRIDE=4 PROBE_COLS=256; else export PROBE_TAPS=2 PROBE_STRIDE=1 PROBE_COLS=1024; fi
      echo "== dbg $d rep $rep conv1 $shape" >> $OUT
      CPC_PLANES_DBG=$d PROBE_TN=0 timeout -k 10 120 python tools/planes_probe.py 20 2>&1 | grep -E "planes nt" >> $OUT
      CPC_PLANES_DBG=$((d + 8)) PROBE_TN=0 timeout -k 10 120 python tools/planes_probe.py 3 2>&1 | grep -E "stamps" | tail -1 >> $OUT
    done
  done
done
cat $OUT
