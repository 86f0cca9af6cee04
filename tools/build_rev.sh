#!/bin/bash
# Builds the library of a git revision as var/libgsr_NAME.so (sources exported to var/src_NAME), for back-to-back
# comparisons with the working tree on one box:  tools/build_rev.sh NAME [REV=HEAD]
# Only meaningful while the revision's buffer layouts (include/gsr.h constants the host code allocates by) equal the
# working tree's: the host code of the working tree drives both libraries.
set -e
name=$1; rev=${2:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$root/var/src_$name"; mkdir -p "$root/var/src_$name"
git -C "$root" archive "$rev" 3d-gaussian-splatting_amd/csrc include | tar -x -C "$root/var/src_$name"
if ! diff <(grep -E "^#define GSR_(REC_FLOATS|STREAM_DWORDS|PAIRGRAD_FLOATS|MAX_VIEWS)" "$root/include/gsr.h") \
          <(grep -E "^#define GSR_(REC_FLOATS|STREAM_DWORDS|PAIRGRAD_FLOATS|MAX_VIEWS)" "$root/var/src_$name/include/gsr.h") > /dev/null; then
  echo "layout constants of $rev differ from the working tree's: not comparable" >&2; exit 1
fi
make -C "$root/var/src_$name/3d-gaussian-splatting_amd/csrc" -j8 > /dev/null 2>&1
cp "$root/var/src_$name/3d-gaussian-splatting_amd/csrc/libgsr.so" "$root/var/libgsr_$name.so"
rm -rf "$root/var/src_$name"
echo "var/libgsr_$name.so"
