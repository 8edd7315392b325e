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
       -Wall -Wno-unused-function -Wno-inline-asm ${BSR_EXTRA_FLAGS:-})
srcs=(bsr_tile bsr_tile_asm bsr_stream bsr_kernels bsr_api bsr_stage bsr_place bsr_comm bsr_engine bsr_refresh)
pids=()
for s in "${srcs[@]}"; do
  [ -f "$here/$s.hip" ] || continue
  # the compiler's per-kernel resource report (registers, scratch) is kept next to the object: tests/test_build_resources.py
  # fails when the tile row pass spills to scratch (it did once, silently, for 4.5 us per launch)
  "$ROCM/bin/hipcc" "${FLAGS[@]}" -Rpass-analysis=kernel-resource-usage -c "$here/$s.hip" -o "$obj/$s.o" 2> "$obj/$s.resources.txt" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
if [ $rc -ne 0 ]; then grep -h -B2 -A6 "error" "$obj"/*.resources.txt >&2 || true; exit 1; fi
grep -h "warning:" "$obj"/*.resources.txt >&2 || true
# scratch (private memory) instructions per function of the tile pass, counted in the disassembly: the resource report's
# ScratchSize also counts stack slots the register allocator reserves and never touches
objdump="$ROCM/lib/llvm/bin/llvm-objdump"
if [ -x "$objdump" ]; then
  tmp="$(mktemp -d)"
  cp "$obj/bsr_tile.o" "$tmp/t.o"
  (cd "$tmp" && "$objdump" --offloading t.o > /dev/null && "$objdump" -d t.o.*gfx950 |
     awk '/^[0-9a-f]+ <.*>:$/ { name = $2; n[name] += 0 } /[ \t]scratch_(load|store)/ { n[name]++ } END { for (k in n) print k, n[k] }') \
    | sort > "$obj/bsr_tile.scratch_ops.txt" || true
  rm -rf "$tmp"
fi
objs=()
for s in "${srcs[@]}"; do [ -f "$obj/$s.o" ] && [ -f "$here/$s.hip" ] && objs+=("$obj/$s.o"); done
"$ROCM/bin/hipcc" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -L"$ROCM/lib" -lrccl -Wl,-rpath,"$ROCM/lib" -o "$out"
echo "built $out"
