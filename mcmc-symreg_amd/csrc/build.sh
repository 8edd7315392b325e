#!/bin/bash
# Builds libbsr_hip.so for gfx950 in-tree (mcmc-symreg_amd/bsr/libbsr_hip.so).
# Every translation unit is compiled on its own (in parallel), then linked; objects live in csrc/build/.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../bsr/libbsr_hip.so"
ROCM="${ROCM_PATH:-/opt/rocm}"
obj="$here/build"
mkdir -p "$obj"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -structurizecfg-skip-uniform-regions=true
       -Wall -Wno-unused-function ${BSR_EXTRA_FLAGS:-})
srcs=(bsr_tile bsr_kernels bsr_api bsr_engine bsr_refresh)
pids=()
for s in "${srcs[@]}"; do
  [ -f "$here/$s.hip" ] || continue
  "$ROCM/bin/hipcc" "${FLAGS[@]}" -c "$here/$s.hip" -o "$obj/$s.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for s in "${srcs[@]}"; do [ -f "$obj/$s.o" ] && [ -f "$here/$s.hip" ] && objs+=("$obj/$s.o"); done
"$ROCM/bin/hipcc" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -L"$ROCM/lib" -lrccl -Wl,-rpath,"$ROCM/lib" -o "$out"
echo "built $out"
