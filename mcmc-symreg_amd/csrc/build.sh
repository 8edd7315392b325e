#!/bin/bash
# Builds libbsr_hip.so for gfx950 in-tree (mcmc-symreg_amd/bsr/libbsr_hip.so).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../bsr/libbsr_hip.so"
ROCM="${ROCM_PATH:-/opt/rocm}"
"$ROCM/bin/hipcc" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -structurizecfg-skip-uniform-regions=true \
  -Wall -Wno-unused-function \
  "$here/bsr_kernels.hip" "$here/bsr_api.hip" "$here/bsr_engine.hip" "$here/bsr_refresh.hip" \
  -L"$ROCM/lib" -lrccl -Wl,-rpath,"$ROCM/lib" -o "$out"
echo "built $out"
