#!/bin/bash
# A/B of tools/variants/* on the GPU (+ the parity suites on the shipped library)
OUT=gpurun_out/${1:-r2e}
mkdir -p $OUT
[ -n "$SKIP_TESTS" ] || python -m pytest tests/test_gpu_parity.py tests/test_gpu_precision.py -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -8 > $OUT/tests.log
rm -f $OUT/variants.txt
CFGS=${CFGS:-f32_split_f16:32 f32_split_f16:64 f16:32 f16:64}
for cfg in $CFGS; do
  prec=${cfg%%:*}; tile=${cfg##*:}
  echo "== $prec tile $tile" >> $OUT/variants.txt
  PREC=$prec TILE=$tile python tools/variants_run.py 2>&1 | grep -v Warning | grep -v amdgpu.ids >> $OUT/variants.txt
done
cat $OUT/tests.log $OUT/variants.txt
