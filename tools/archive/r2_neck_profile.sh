#!/bin/bash
# neck part of tools/profile_round.sh (after the row-window conv kernel became the default)
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
NECK="python $ROOT/tools/neck_bench.py 16 40"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/neck_trace -o trace -- $NECK > $OUT/neck_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/neck_pmc_fetch -o pmc -- $NECK > $OUT/neck_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/neck_pmc_write -o pmc -- $NECK > $OUT/neck_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/neck_pmc_lds -o pmc -- $NECK > $OUT/neck_pmc_lds.log 2>&1
cd $ROOT
db=$(find $OUT/neck_trace -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $OUT/neck_trace_kernel_stats.csv > /dev/null
for d in neck_pmc_fetch neck_pmc_write neck_pmc_lds; do
  db=$(find $OUT/$d -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py $db $OUT/$d.csv > /dev/null
done
find $OUT -name "*.db" -delete
grep -h neck $OUT/neck_trace_kernel_stats.csv $OUT/neck_pmc_fetch.csv $OUT/neck_pmc_write.csv $OUT/neck_pmc_lds.csv
tail -1 $OUT/neck_trace.log
