#!/bin/bash
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_precision.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r2b/precision_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench_default.json 2> gpurun_out/r2b/bench_default.err
python bench.py --steps 20 --warmup 5 --precision bf16 --no-e2e > gpurun_out/r2b/bench_bf16.json 2> gpurun_out/r2b/bench_bf16.err
OETR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-e2e > gpurun_out/r2b/bench_gloo2.json 2> gpurun_out/r2b/bench_gloo2.err
python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2b/bench_refuse.json 2> gpurun_out/r2b/bench_refuse.err; echo "rc=$?" >> gpurun_out/r2b/bench_refuse.err
for p in f32_split_f16 bf16; do
  python tools/phase_timing.py $p > gpurun_out/r2b/phase32_$p.txt 2>&1
  python tools/phase_timing64.py $p > gpurun_out/r2b/phase64_$p.txt 2>&1
done
cat gpurun_out/r2b/precision_tests.log; cat gpurun_out/r2b/phase32_*.txt gpurun_out/r2b/phase64_*.txt | grep -v Warning
