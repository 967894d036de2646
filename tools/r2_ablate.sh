#!/bin/bash
mkdir -p gpurun_out/r2g
for cfg in f32_split_f16:32 f32_split_f16:64 bf16:32; do
  prec=${cfg%%:*}; tile=${cfg##*:}
  ABL_CUM=1 ABL_PREC=$prec ABL_TILE=$tile python tools/ablate_run.py 2>&1 | grep -v Warning | grep -v amdgpu.ids > gpurun_out/r2g/cum_${prec}_$tile.txt
done
cat gpurun_out/r2g/cum_*.txt
