#!/bin/bash
# per-shape table of the plane-fed launches, one-tap-per-stage form (CPC_PLANES_NO_PAIR=1) and pair form, same box
for v in nopair pair; do
  if [ $v = nopair ]; then export CPC_PLANES_NO_PAIR=1; else unset CPC_PLANES_NO_PAIR; fi
  bash tools/trace_step.sh r04s_$v --also "" || exit 1
  CPC_PLANES_DBG=8 timeout -k 10 200 python3 bench.py --steps 2 --warmup 2 --cpu-seconds 0 --no-prof --also "" > gpurun_out/r04s_${v}_stamps.json 2> gpurun_out/r04s_${v}_stamps.log
  python3 tools/planes_shapes.py gpurun_out/kt_r04s_$v gpurun_out/r04s_${v}_stamps.log > gpurun_out/r04s_${v}_shapes.md 2>&1
done
cat gpurun_out/r04s_nopair_shapes.md gpurun_out/r04s_pair_shapes.md
