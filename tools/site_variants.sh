#!/bin/bash
# Per-GEMM-site precision study: one library per (site, reduction) with ONE site of the 64-token
# encoder kernels reduced (common.h: OETR_SITE_* / SITE_*), everything else fp32-class.
#   tools/site_variants.sh build      (CPU, needs the shipped objects in csrc/: make first)
#   tools/site_variants.sh run OUT    (GPU: drift of every variant on the goldens -> OUT json lines)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/imagematching_oetr_amd/csrc
VDIR=$ROOT/tools/variants
SITES="Q K V MERGE MLP1 MLP2 DEC_K DEC_V"
if [ "$1" = build ]; then
  n=0
  for site in $SITES; do for red in 1 2 3; do
    name=site_${site}_${red}
    mkdir -p $VDIR/$name
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DOETR_SITE_${site}=${red} -c $CSRC/encoder.hip -o $VDIR/$name/encoder.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $VDIR/$name/liboetr_hip.so $VDIR/$name/encoder.o \
        $CSRC/api.o $CSRC/decoder.o $CSRC/heads.o $CSRC/attention.o $CSRC/neck.o $CSRC/crop.o $CSRC/reader.o ) &
    n=$((n+1)); if [ $((n % 6)) = 0 ]; then wait; fi
  done; done
  wait
  ls $VDIR
else
  OUT=${2:-gpurun_out/r3_site_drift.jsonl}
  mkdir -p $(dirname $OUT); : > $OUT
  python $ROOT/tools/policy_check.py f32_split_f16@64 f32_split_qk16 f16@64 bf16@64 2>/dev/null | grep '^{' | sed 's/^{/{"variant": "shipped", /' >> $OUT
  for site in $SITES; do for red in 1 2 3; do
    name=site_${site}_${red}
    [ -f $VDIR/$name/liboetr_hip.so ] || continue
    OETR_HIP_LIB=$VDIR/$name/liboetr_hip.so python $ROOT/tools/policy_check.py f32_split_f16@64 2>/dev/null | grep '^{' \
      | sed "s/^{/{\"variant\": \"$name\", \"site\": \"$site\", \"reduction\": $red, /" >> $OUT
  done; done
  cat $OUT | cut -c1-260
fi
