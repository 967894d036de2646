B="python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-exact-f32 --no-other-configs --no-trace --no-power"
P="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['ms_per_step'], d['serial']['pairs_per_s'])"
for i in 1 2; do
OETR_BENCH_FORCE_PG=1 $B 2>/dev/null | grep '^{' | python -c "$P" "pg steps20"
$B 2>/dev/null | grep '^{' | python -c "$P" "nopg steps20"
done
