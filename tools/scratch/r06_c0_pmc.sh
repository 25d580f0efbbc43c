mkdir -p gpurun_out
export TMPDIR=/tmp
for f in ${FORMS:-3 0}; do
  export CPC_CONV0_BWD=$f
  bash tools/pmc_kernels.sh c0f$f "conv0_bwd" small > gpurun_out/c0_pmc_$f.txt 2>&1 || { echo "pmc $f failed"; tail gpurun_out/c0_pmc_$f.txt; exit 1; }
  cat gpurun_out/c0_pmc_$f.txt
done
