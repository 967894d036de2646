#!/bin/bash
# L1/L2 request counters for conv kernel variants (tools/variants/<name>)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/neck_pmc2
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*" | sort -u > $OUT/counters.txt
for v in "$@"; do
  CMD="python $ROOT/tools/neck_variants.py $v"
  timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/a_$v -o pmc -- $CMD > $OUT/a_$v.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum -d $OUT/b_$v -o pmc -- $CMD > $OUT/b_$v.log 2>&1
done
cd $ROOT
for v in "$@"; do for p in a b; do
  db=$(find $OUT/${p}_$v -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py $db $OUT/${p}_$v.csv > /dev/null
  echo "== $p $v"; head -1 $OUT/${p}_$v.csv; grep conv $OUT/${p}_$v.csv; tail -3 $OUT/${p}_$v.log | head -2
done; done
find $OUT -name "*.db" -delete
wc -l $OUT/counters.txt
