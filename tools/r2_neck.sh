#!/bin/bash
mkdir -p gpurun_out/r2l
for v in base slicemajor; do
  echo "== $v" >> gpurun_out/r2l/neck.txt
  OETR_HIP_LIB=tools/variants/$v/liboetr_hip.so python tools/neck_bench.py 16 40 2>&1 | grep -v Warning | grep -v amdgpu >> gpurun_out/r2l/neck.txt
  OETR_HIP_LIB=tools/variants/$v/liboetr_hip.so python tools/neck_bench.py 64 40 2>&1 | grep -v Warning | grep -v amdgpu >> gpurun_out/r2l/neck.txt
done
cat gpurun_out/r2l/neck.txt
