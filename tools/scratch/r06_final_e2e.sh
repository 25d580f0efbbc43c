#!/bin/bash
# end-to-end runs of the round's last build on the reference's audio fixture + soak
mkdir -p gpurun_out
timeout -k 10 400 python tools/fixture_parity.py 4 2e-4 default > gpurun_out/r06_fixture_parity_36steps.log 2>&1 || { echo "fixture parity failed"; tail -5 gpurun_out/r06_fixture_parity_36steps.log; exit 1; }
tail -3 gpurun_out/r06_fixture_parity_36steps.log
timeout -k 10 500 python tools/train_fixture.py 12 > gpurun_out/r06_fixture_training.log 2>&1 || { echo "fixture training failed"; tail -5 gpurun_out/r06_fixture_training.log; exit 1; }
tail -3 gpurun_out/r06_fixture_training.log
bash tools/scratch/r06_soak.sh > gpurun_out/r06_soak_stdout.txt 2>&1
tail -25 gpurun_out/r06_soak.txt
