# bench.py with / without a process group (RCCL at world 1), by HIP hardware-queue count (GPU_MAX_HW_QUEUES) and stream count
for q in 4 8; do
export GPU_MAX_HW_QUEUES=$q
for s in 3 4; do
OETR_BENCH_FORCE_PG=1 python bench.py --steps 50 --warmup 5 --repeats 3 --no-e2e --no-cpu-baseline --no-exact-f32 --no-other-configs --no-trace --streams $s 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues', $q, 'pg streams', $s, d['value'], d['ms_per_step'], d['serial']['pairs_per_s'])"
python bench.py --steps 50 --warmup 5 --repeats 3 --no-e2e --no-cpu-baseline --no-exact-f32 --no-other-configs --no-trace --streams $s 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues', $q, 'nopg streams', $s, d['value'], d['ms_per_step'], d['serial']['pairs_per_s'])"
done; done
