#!/bin/bash
# Determinism soak of the round-4 tree (tools/determinism_hunt.py: interleaved shapes, every stage output compared with the
# first run of its shape): the shipped rules, then the decoder's workgroups per image and the tail form pinned, then the
# workspace behind the status block overwritten with random numbers before every run.
#   tools/r4_soak.sh [seconds per mode] > profiles/r4_determinism_soak2.txt
B=${1:-60}
run() { echo "== $*"; env "${@:3}" timeout 400 python tools/determinism_hunt.py $1 $2 $B 2>&1 | grep -v amdgpu.ids | tail -1; }
for cfg in "f32_split_f16 64" "f32_split_f16 32" "f32_split_qk16 64" "f32 32"; do run $cfg; done
for k in 1 4; do run f32_split_f16 64 HUNT_DECSPLIT=$k; run f32_split_f16 32 HUNT_DECSPLIT=$k; done
run f32 32 HUNT_DECSPLIT=4
for t in 1 2 3; do run f32_split_f16 64 HUNT_TAILMODE=$t; done
run f32_split_f16 64 HUNT_TAILMODE=2 HUNT_DECSPLIT=4
run f32_split_f16 64 HUNT_FILL=rand
run f32_split_f16 32 HUNT_FILL=nan HUNT_DECSPLIT=4
