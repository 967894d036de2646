#!/bin/bash
mkdir -p gpurun_out/r2p; : > gpurun_out/r2p/sweep.txt
for tile in 32 64; do for st in 2 3 4 6; do
  python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-e2e --no-trace --no-exact-f32 --streams $st --enc-tile $tile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $tile streams $st', d['value'], d['ms_per_step'], d['timing'])" >> gpurun_out/r2p/sweep.txt
done; done
cat gpurun_out/r2p/sweep.txt
