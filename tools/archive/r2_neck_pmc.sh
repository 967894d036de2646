#!/bin/bash
# stall breakdown of the PatchMerging conv kernels (gather vs row window)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/neck_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for kind in row_window gather; do
  NECK="python $ROOT/tools/neck_bench.py 16 40 $kind"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/sq_$kind -o pmc -- $NECK > $OUT/sq_$kind.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/sq2_$kind -o pmc -- $NECK > $OUT/sq2_$kind.log 2>&1
done
cd $ROOT
for d in sq_row_window sq_gather sq2_row_window sq2_gather; do
  db=$(find $OUT/$d -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py $db $OUT/$d.csv > /dev/null
  grep conv $OUT/$d.csv; head -1 $OUT/$d.csv
done
find $OUT -name "*.db" -delete
