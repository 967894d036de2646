OUT=gpurun_out/r5; mkdir -p $OUT
OETR_BENCH_FORCE_PG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-exact-f32 2>> $OUT/bench.err | grep '^{' > $OUT/bench_rccl_world1.json
OETR_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-e2e --no-exact-f32 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gloo2.json
OETR_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --workload mixed --steps 10 --warmup 3 --precision f32_split_qk16 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gloo2_mixed.json
OETR_BENCH_FORCE_PG=1 timeout 300 python bench.py --workload mixed --steps 10 --warmup 3 --precision f32_split_qk16 2>> $OUT/bench.err | grep '^{' > $OUT/bench_rccl_world1_mixed.json
for f in bench_rccl_world1 bench_gloo2 bench_gloo2_mixed bench_rccl_world1_mixed; do python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[1], d['value'], d['n_gpus'], d.get('serial', {}).get('pairs_per_s'), json.dumps(d.get('process_group'))[:200])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
tail -5 $OUT/bench.err
