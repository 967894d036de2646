#!/bin/bash
# full GPU suite + default bench + full-attention bench
OUT=gpurun_out/${1:-r2j}
mkdir -p $OUT
python -m pytest tests -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -12 > $OUT/tests.log
python bench.py --steps 20 --warmup 5 2>$OUT/bench.err > $OUT/bench.json
for L in 1024 4096; do
  python bench.py --kernel full_attention --L $L --steps 20 --warmup 3 --repeats 5 2>/dev/null > $OUT/full_attention_L$L.json
done
cat $OUT/tests.log; python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','timing','serial','end_to_end_pairs_per_s','neck_kernels_us') if k in d})
print(d.get('roofline')); print(d.get('exact_f32'))
for L in (1024,4096):
    f=json.load(open('$OUT/full_attention_L%d.json'%L)); print(L, f['variants'])
PY
