#!/bin/bash
# tools/variants/<name>/liboetr_hip.so with encoder.hip alone rebuilt under extra flags (the other objects are the
# in-tree ones: run `make -C imagematching_oetr_amd/csrc` first) - one-process A/Bs with tools/variants_run.py.
# usage: tools/enc_variants.sh base "" ring5 "-DOETR_RING2=5" ...
set -e
cd "$(dirname "$0")/../imagematching_oetr_amd/csrc"
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  OUT=../../tools/variants/$name
  mkdir -p $OUT
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c encoder.hip -o $OUT/encoder.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liboetr_hip.so $OUT/encoder.o api.o decoder.o heads.o attention.o neck.o crop.o reader.o calib.o &&
    rm -f $OUT/encoder.o ) &
done
wait
