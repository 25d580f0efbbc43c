#!/bin/bash
# the pair form of the plane-fed NT kernel against the one-tap-per-stage form: probe shapes (checked against fp64), then the step
OUT=gpurun_out/ab_pair.txt
: > $OUT
for rep in 1 2; do
  for v in pair nopair; do
    if [ $v = nopair ]; then export CPC_PLANES_NO_PAIR=1; else unset CPC_PLANES_NO_PAIR; fi
    for shape in ${SHAPES:-fwd1 dgrad1 fwd2 dgrad3}; do
      case $shape in
        fwd1) export PROBE_TAPS=8 PROBE_STRIDE=4 PROBE_COLS=256 PROBE_L=1024;;
        dgrad1) export PROBE_TAPS=2 PROBE_STRIDE=1 PROBE_COLS=1024 PROBE_L=1024;;
        fwd2) export PROBE_TAPS=4 PROBE_STRIDE=2 PROBE_COLS=256 PROBE_L=512;;
        dgrad3) export PROBE_TAPS=2 PROBE_STRIDE=1 PROBE_COLS=512 PROBE_L=256;;
      esac
      echo "== $v rep $rep $shape" >> $OUT
      PROBE_TN=0 timeout -k 10 120 python tools/planes_probe.py 20 2>&1 | grep -E "planes nt|fp64" >> $OUT
    done
  done
done
unset CPC_PLANES_NO_PAIR
cat $OUT
