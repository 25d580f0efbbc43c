set -e
mkdir -p gpurun_out/ab
for cfg in large small; do
for v in ${VARIANTS:-grusv0 grusv1 grusv2}; do
  export CPC2_HIP_LIB=$PWD/tools/variant/lib_$v.so
  python bench.py --config $cfg --cpu-seconds 0 --no-prof --also "" --steps 8 --warmup 4 > gpurun_out/ab/st_$cfg$v.json 2>gpurun_out/ab/st_$cfg$v.err || true
  echo "== $cfg $v"; grep "gru fwd stamps" gpurun_out/ab/st_$cfg$v.err | tail -2 | cut -c1-260
done; done
