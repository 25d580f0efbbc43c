#!/bin/bash
# Diagnostic build of libcpc2_hip.so with extra compiler flags on ONE source, or on every .hip source ("all"); the other
# objects are taken from build/:
#   tools/build_variant.sh encoder.hip -DCPC_C0_DBG   ->  tools/variant/libcpc2_hip.so    (use with CPC2_HIP_LIB=...)
#   tools/build_variant.sh all -Xclang -target-feature -Xclang -packed-fp32-ops
set -e
cd "$(dirname "$0")/.."
src=$1; shift
mkdir -p build/variant tools/variant
python -c 'import __graft_entry__ as g; g.build()' >/dev/null
rm -f build/variant/*.o
if [ "$src" = all ]; then srcs=$(cd cpc2_amd/csrc && ls *.hip); else srcs=$src; fi
for s in $srcs; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -pthread --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c cpc2_amd/csrc/$s -o build/variant/$s.o 2>&1 | grep -v "not a recognized feature" || true &
done
wait
objs=""
for o in build/*.o; do b=$(basename $o); if [ -f build/variant/$b ]; then objs="$objs build/variant/$b"; else objs="$objs $o"; fi; done
out=${VARIANT_OUT:-tools/variant/libcpc2_hip.so}          # (tools/variant/ does not travel to the GPU box: .gpurunignore)
mkdir -p "$(dirname "$out")"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out" $objs
echo "$out"
