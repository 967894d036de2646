#!/bin/bash
mkdir -p gpurun_out/r2o
python -m pytest tests/test_gpu_neck.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -3 > gpurun_out/r2o/neck.txt
echo "== shipped" >> gpurun_out/r2o/neck.txt
python tools/neck_rows.py 2>&1 | grep -v Warning | grep -v amdgpu >> gpurun_out/r2o/neck.txt
cat gpurun_out/r2o/neck.txt
