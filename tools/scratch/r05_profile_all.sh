#!/bin/bash
# round-5 evidence run: the bench line, kernel stats + PMC passes (small, large, transformer), step timeline, per-shape stamps
set -u
TAG=${1:-r05a}
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
bash tools/gpu_profile.sh $TAG small > gpurun_out/prof_$TAG.log 2>&1 || { echo "profile small failed"; tail -5 gpurun_out/prof_$TAG.log; exit 1; }
bash tools/gpu_profile.sh ${TAG}_large large > gpurun_out/prof_${TAG}_large.log 2>&1 || { echo "profile large failed"; exit 1; }
bash tools/trace_step.sh $TAG --also "" || exit 1
python3 tools/show_trace.py gpurun_out/kt_$TAG > gpurun_out/${TAG}_step_timeline.txt 2>&1
bash tools/trace_step.sh ${TAG}_large --config large --also "" || exit 1
python3 tools/show_trace.py gpurun_out/kt_${TAG}_large > gpurun_out/${TAG}_large_step_timeline.txt 2>&1
bash tools/trace_step.sh ${TAG}_transformer --config transformer --also "" || exit 1
python3 tools/show_trace.py gpurun_out/kt_${TAG}_transformer > gpurun_out/${TAG}_transformer_step_timeline.txt 2>&1
CPC_PLANES_DBG=8 timeout -k 10 200 python3 bench.py --steps 2 --warmup 2 --cpu-seconds 0 --no-prof --also "" > gpurun_out/${TAG}_stamps.json 2> gpurun_out/${TAG}_stamps.log
python3 tools/planes_shapes.py gpurun_out/kt_$TAG gpurun_out/${TAG}_stamps.log > gpurun_out/${TAG}_planes_shapes.md 2>&1
cat gpurun_out/${TAG}_planes_shapes.md | head -20
tail -3 gpurun_out/${TAG}_step_timeline.txt | head -2
