B="python bench.py --steps 96 --warmup 8 --repeats 3 --no-e2e --no-cpu-baseline --no-exact-f32 --no-other-configs --no-trace"
P="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['ms_per_step'], d['serial']['pairs_per_s'])"
for g in auto 1 0; do
OETR_BENCH_FORCE_PG=1 $B --gather-on-stream $g 2>/dev/null | grep '^{' | python -c "$P" "pg on_stream=$g"
done
$B 2>/dev/null | grep '^{' | python -c "$P" "nopg default"
