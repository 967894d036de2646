#!/bin/bash
# tools/variants/<name>/liboetr_hip.so for the write-through store A/B (tools/wt_ab.py): only encoder.hip
# depends on OETR_WT, the other objects are the in-tree ones (run `make -C imagematching_oetr_amd/csrc` first).
# usage: tools/wt_variants.sh base 0 wt1 1 wt2 2 wt4 4 wt7 7
set -e
cd "$(dirname "$0")/../imagematching_oetr_amd/csrc"
while [ $# -gt 0 ]; do
  name=$1; mask=$2; shift 2
  OUT=../../tools/variants/$name
  mkdir -p $OUT
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DOETR_WT=$mask -c encoder.hip -o $OUT/encoder.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liboetr_hip.so $OUT/encoder.o api.o decoder.o heads.o attention.o neck.o crop.o reader.o &&
    rm -f $OUT/encoder.o ) &
done
wait
