#!/bin/bash
# Round-4 hazard study: variants of encoder.hip only (the other objects come from the main build).
# usage: tools/r4_hazard_build.sh name "-Dflags" [name "-Dflags" ...]   -> tools/variants/<name>/liboetr_hip.so
set -e
cd "$(dirname "$0")/../imagematching_oetr_amd/csrc"
[ -f api.o ] || make -j8 >/dev/null
pids=()
names=()
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  OUT=../../tools/variants/$name; mkdir -p $OUT
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c encoder.hip -o $OUT/encoder.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liboetr_hip.so api.o $OUT/encoder.o decoder.o heads.o attention.o neck.o crop.o reader.o &&
    rm -f $OUT/encoder.o && echo "built $name" ) &
  if [ $(jobs -r | wc -l) -ge 8 ]; then wait -n; fi
done
wait
