#!/bin/bash
# round-2 first GPU pass: new precision/range tests, old suites, quick per-precision timings
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_precision.py -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r2a/precision_tests.log
python -m pytest tests/test_gpu_precision.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r2a/precision_tests_all.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_neck.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r2a/old_tests.log
for p in f32_split_f16 f16 bf16; do
  python bench.py --precision $p --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/r2a/bench_$p.json 2> gpurun_out/r2a/bench_$p.err
done
cat gpurun_out/r2a/precision_tests.log
tail -5 gpurun_out/r2a/old_tests.log
