#!/bin/bash
# Register / LDS / scratch figures of every kernel in the shipped sources, one line per kernel:
#   tools/kernel_resources.sh > profiles/<round>_kernel_resources.txt
# (hipcc's -Rpass-analysis=kernel-resource-usage remarks, same flags as csrc/Makefile; no GPU needed)
set -e
cd "$(dirname "$0")/../imagematching_oetr_amd/csrc"
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage of the shipped sources ($(git log -1 --format=%h 2>/dev/null || echo tree)), one line per kernel"
echo "# VGPRs | AGPRs | scratch B/lane | VGPR spills | SGPR spills | LDS B/block | waves/SIMD | kernel"
for f in encoder decoder heads attention neck crop reader; do
  extra=""; case $f in encoder|attention) extra="-fno-slp-vectorize";; esac      # (per-file flags of csrc/Makefile)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -Rpass-analysis=kernel-resource-usage -c $f.hip -o /dev/null 2>&1 |
    python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
    else:
        cur[k] = v
    if k == "LDS Size [bytes/block]":
        print("%4s | %3s | %4s | %3s | %3s | %7s | %2s | %s" % (cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("VGPRs Spill"), cur.get("SGPRs Spill"), cur.get("LDS Size [bytes/block]"), cur.get("Occupancy [waves/SIMD]"), cur["name"]))
'
done
