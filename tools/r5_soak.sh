#!/bin/bash
# Determinism soak of the round-5 tree (split-f16 linear-attention state in ONE code path):
# the shipped build in the precision / tile modes, the MASKED instantiations, forced pre-reduction, poisoned
# workspaces - then the two amplifier builds of round 4's hazard study (tools/enc_variants.sh:
# amp_vmcnt = -DOETR_SOAK_AMP=1, amp_prio = -DOETR_SOAK_AMP=2, amp_both = 3), under which the two-path
# forms of that state failed 14 .. 8 002 of 37 000.
#   tools/r5_soak.sh [seconds per mode] > profiles/r5_determinism_soak.txt
B=${1:-60}
run() { echo "== $*"; env "${@:4}" timeout 600 python tools/determinism_hunt.py $1 $2 $B $3 2>&1 | grep -v amdgpu.ids | tail -1; }
for v in "" amp_vmcnt amp_prio amp_both; do
  [ -n "$v" ] && [ ! -f tools/variants/$v/liboetr_hip.so ] && continue
  echo "#### library: ${v:-shipped}"
  for cfg in "f32_split_f16 64" "f32_split_f16 32" "f32_split_qk16 64" "f32_split_qk16 32"; do run $cfg "$v"; done
  run f32_split_f16 64 "$v" HUNT_MASKS=1
  run f32_split_f16 32 "$v" HUNT_MASKS=1
  run f32_split_f16 64 "$v" HUNT_PREREDUCE=1
  run f32_split_f16 64 "$v" HUNT_TAILMODE=2 HUNT_DECSPLIT=4
  run f32_split_f16 32 "$v" HUNT_FILL=rand
  run f32_split_f16 64 "$v" HUNT_THRASH_MB=512
done
