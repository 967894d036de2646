#!/bin/bash
# Builds the gate-1 prototype library (gfx950).  Usage: tools/chain128/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
  chain128.hip -o libchain128.so
