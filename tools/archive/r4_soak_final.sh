#!/bin/bash
# Determinism soak of the FINAL round-4 tree (write-through stores, masks): the shipped rules in the four precision / tile
# modes, then the MASKED kernel instantiations (forward_dummy's masks with holes on every shape), then a poisoned workspace.
#   tools/r4_soak_final.sh [seconds per mode] > profiles/r4_determinism_soak3.txt
B=${1:-60}
run() { echo "== $*"; env "${@:3}" timeout 400 python tools/determinism_hunt.py $1 $2 $B 2>&1 | grep -v amdgpu.ids | tail -1; }
for cfg in "f32_split_f16 64" "f32_split_f16 32" "f32_split_qk16 64" "f32 32"; do run $cfg; done
for cfg in "f32_split_f16 64" "f32_split_f16 32" "f32 32"; do run $cfg HUNT_MASKS=1; done
run f32_split_f16 64 HUNT_MASKS=1 HUNT_TAILMODE=2 HUNT_DECSPLIT=4
run f32_split_f16 64 HUNT_FILL=rand
run f32_split_f16 32 HUNT_FILL=nan HUNT_MASKS=1
