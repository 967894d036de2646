#!/bin/bash
# Other BASELINE / SURVEY 8(d) shapes through bench.py (hot path only, no CPU leg): one JSON
# line each -> gpurun_out/<tag>_extra_configs.jsonl
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_extra_configs.jsonl
mkdir -p $ROOT/gpurun_out; : > $OUT
run() { timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-e2e "$@" >> $OUT 2>/dev/null; }
run --pairs-per-gpu 1 --steps 200                                  # configs[0] shape: 1 pair @640
run --pairs-per-gpu 64 --steps 40 --warmup 5                       # config 3: 64 pairs @640
run --pairs-per-gpu 32 --size 1024 --steps 20 --warmup 3           # config 4: 32 pairs @1024 (32x32 tokens)
run --pairs-per-gpu 32 --size 1024 --steps 20 --warmup 3 --enc-tile 32   # config 4, LDS-tile sweep: 32-row encoder tiles
run --pairs-per-gpu 32 --size 1024 --steps 20 --warmup 3 --enc-tile 64   #                            64-row (what auto picks)
run --pairs-per-gpu 4 --size 2048 --steps 10 --warmup 2            # config 4b: 64x64 tokens
run --pairs-per-gpu 8 --size 640 --size2 1280 --steps 50 --warmup 5  # config 5: L1=400 vs L2=1600
run --pairs-per-gpu 8 --enc-tile 32 --streams 2                    # tile sweep at configs[1]: 32-row tiles
run --pairs-per-gpu 8 --precision f32 --streams 2                  # exact-f32 MFMA mode
python - <<PY
import json
for l in open("$OUT"):
    d = json.loads(l)
    print(d['config']['workload'][:70], '|', d['config']['streams'], 'streams tile', d['config']['encoder_tile_rows'], '|', d['value'], 'pairs/s', d['ms_per_step'], 'ms | serial', d['serial']['pairs_per_s'])
PY
