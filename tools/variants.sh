#!/bin/bash
# Build timing variants of the library: tools/variants/<name>/liboetr_hip.so
# usage: tools/variants.sh name "-DFLAG1 -DFLAG2" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/../imagematching_oetr_amd/csrc"
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  OUT=../../tools/variants/$name
  mkdir -p $OUT
  for f in api encoder decoder heads attention neck crop reader calib; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $f.hip -o $OUT/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liboetr_hip.so $OUT/*.o
  rm -f $OUT/*.o
done
