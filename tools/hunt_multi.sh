#!/bin/bash
# multi-shape determinism hunt of library variants: usage tools/hunt_multi.sh "<variants>" [budget_s] ["prec tile" ...]
V=${1:-cur}; B=${2:-25}; shift 2
[ $# -eq 0 ] && set -- "f16 64" "f32_split_f16 64"
for v in $V; do
  for cfg in "$@"; do
    echo "== $v $cfg"; HUNT_VERBOSE=${HUNT_VERBOSE:-0} timeout 300 python tools/determinism_hunt.py $cfg $B $v 2>&1 | grep -v amdgpu.ids | tail -${HUNT_TAIL:-1}
  done
done
