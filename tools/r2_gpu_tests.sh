#!/bin/bash
# usage: tools/r2_gpu_tests.sh <outdir> [pytest args]
OUT=gpurun_out/${1:-r2t}; shift
mkdir -p $OUT
python -m pytest "${@:-tests}" -q -m gpu --no-header -p no:cacheprovider -s 2>&1 | grep -v amdgpu.ids | tail -40 > $OUT/tests.log
cat $OUT/tests.log
