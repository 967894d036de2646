#!/bin/bash
# Determinism soak of the round-6 tree: the shipped library and the amplified one that `make` builds beside it
# (liboetr_hip_soak.so, -DOETR_SOAK_AMP=3: vmcnt(0) before every GEMM step + s_setprio 3 around the state - the
# amplifiers under which the two-path forms of the split-f16 KV state failed 14 .. 8 002 of 37 000),
# in the precision / tile modes, the MASKED instantiations, forced pre-reduction, the direct tail with the split
# decoder, poisoned workspaces, cold L2.  tools/determinism_hunt.py does the comparing (every stage output and
# workspace buffer of every forward against the first of its shape).
#   tools/r6_soak.sh [seconds per mode] > profiles/r6_determinism_soak.txt
B=${1:-40}
cd "$(dirname "$0")/.."
mkdir -p tools/variants/amp_both
cp imagematching_oetr_amd/csrc/liboetr_hip_soak.so tools/variants/amp_both/liboetr_hip.so
run() { echo "== $*"; env "${@:4}" timeout 600 python tools/determinism_hunt.py $1 $2 $B $3 2>&1 | grep -v amdgpu.ids | tail -1; }
for v in "" amp_both; do
  echo "#### library: ${v:-shipped}"
  for cfg in "f32_split_f16 64" "f32_split_f16 32" "f32_split_qk16 64" "f32_split_qk16 32"; do run $cfg "$v"; done
  run f32_split_f16 64 "$v" HUNT_MASKS=1
  run f32_split_f16 32 "$v" HUNT_MASKS=1
  run f32_split_f16 64 "$v" HUNT_PREREDUCE=1
  run f32_split_f16 64 "$v" HUNT_TAILMODE=2 HUNT_DECSPLIT=4
  run f32_split_f16 32 "$v" HUNT_FILL=rand
  run f32_split_f16 64 "$v" HUNT_THRASH_MB=512
  run f32_split_f16 32 "$v" HUNT_ATTENTION=full
  run f32 32 "$v"
done
rm -rf tools/variants/amp_both
