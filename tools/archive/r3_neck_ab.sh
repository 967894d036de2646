#!/bin/bash
# time + FETCH_SIZE of the neck conv kernel per work-item order
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT
for v in neck_tile neck_slice; do
  export OETR_HIP_LIB=$ROOT/tools/variants/$v/liboetr_hip.so
  python $ROOT/tools/neck_bench.py 16 40 2>/dev/null | tail -1 | sed "s/^/$v: /" >> $OUT/neck_ab.txt
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_$v -o pmc -- python $ROOT/tools/neck_bench.py 16 40 > /dev/null 2>&1
  cd $ROOT; db=$(find $OUT/pmc_$v -name "*.db" | head -1); python tools/rocpd_pmc.py $db $OUT/pmc_$v.csv > /dev/null; grep -i "neck" $OUT/pmc_$v.csv | sed "s/^/$v: /" >> $OUT/neck_ab.txt
  find $OUT/pmc_$v -name "*.db" -delete
done
cat $OUT/neck_ab.txt
