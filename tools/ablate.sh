#!/bin/bash
# Builds tools/ablate/liboetr_hip.so with -DOETR_ABLATE (timing attribution only).
set -e
cd "$(dirname "$0")/../imagematching_oetr_amd/csrc"
OUT=../../tools/ablate
mkdir -p $OUT
for f in api encoder decoder heads attention neck crop; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DOETR_ABLATE -c $f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liboetr_hip.so $OUT/*.o
rm -f $OUT/*.o
