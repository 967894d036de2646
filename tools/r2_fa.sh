#!/bin/bash
mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_parity.py -q -m gpu --no-header -p no:cacheprovider -k "attention" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r2i/tests.log
for L in 1024 4096; do
  python bench.py --kernel full_attention --L $L --steps 20 --warmup 3 --repeats 5 2>/dev/null > gpurun_out/r2i/full_attention_L$L.json
done
cat gpurun_out/r2i/tests.log gpurun_out/r2i/full_attention_L*.json
