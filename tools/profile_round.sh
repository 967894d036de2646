#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC passes.
# usage: tools/profile_round.sh <tag>     outputs under gpurun_out/<tag>/
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-e2e --no-trace --streams 1 --steps 50 --warmup 5"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
# PMC in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o pmc -- $BENCH > $OUT/pmc_lds.log 2>&1
# the default (value) mode of bench.py: batches overlapped on 3 streams, 64-row encoder tiles
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/overlap_trace -o trace -- python $ROOT/bench.py --no-cpu-baseline --no-e2e --no-trace --steps 50 --warmup 5 > $OUT/overlap_trace.log 2>&1
# neck: 16 backbone maps of 40x40 through the HIP neck
NECK="python $ROOT/tools/neck_bench.py 16 40"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/neck_trace -o trace -- $NECK > $OUT/neck_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/neck_pmc_fetch -o pmc -- $NECK > $OUT/neck_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/neck_pmc_write -o pmc -- $NECK > $OUT/neck_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/neck_pmc_lds -o pmc -- $NECK > $OUT/neck_pmc_lds.log 2>&1
cd $ROOT
# summarise on the box; the raw rocpd databases are too big to carry back
for d in trace overlap_trace neck_trace; do
  db=$(find $OUT/$d -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $OUT/${d}_kernel_stats.csv > /dev/null
done
for d in pmc_fetch pmc_write pmc_sq pmc_lds neck_pmc_fetch neck_pmc_write neck_pmc_lds; do
  db=$(find $OUT/$d -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py $db $OUT/$d.csv > /dev/null
done
find $OUT -name "*.db" -delete
cat $OUT/bench.json
ls $OUT
