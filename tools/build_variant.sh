#!/bin/bash
# Builds a variant of the library for back-to-back comparisons on one box (tools/bench_variants.sh):
#   tools/build_variant.sh NAME "-DGSR_FOO=1 ..." [file.hip ...]   ->  var/libgsr_NAME.so
# The named sources (default: composite.hip) are compiled with the extra flags, the other objects are the in-tree ones.
# A variant is tied to the tree it was built from: after a change of a buffer layout (include/gsr.h constants the host
# code allocates by) delete var/ - a stale library writes rows of the old width into buffers of the new one.
set -e
name=$1; flags=$2; shift 2 || true
files=${@:-composite.hip}
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/3d-gaussian-splatting_amd/csrc
mkdir -p "$root/var/$name"
make -C "$csrc" -j8 > /dev/null
objs=""
for src in api preprocess scan_sort binning blocklists composite loss adam densify knn; do
  if echo " $files " | grep -q " $src.hip "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize \
      -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-sanitize -Wall -Wno-unused-function $flags \
      -c "$csrc/$src.hip" -o "$root/var/$name/$src.o"
    objs="$objs $root/var/$name/$src.o"
  else
    objs="$objs $csrc/$src.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$root/var/libgsr_$name.so" $objs
echo "var/libgsr_$name.so"
