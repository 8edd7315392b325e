#!/bin/bash
# Builds libbsr_hip.so for gfx950 in-tree (mcmc-symreg_amd/bsr/libbsr_hip.so): the SHIPPED library, one kernel per situation.
# `build.sh variants` builds mcmc-symreg_amd/bsr/libbsr_hip_variants.so as well: the same sources with -DBSR_TEST_VARIANTS -- every
# interpreter of the streaming pass (BSR_STREAM_ASM=0..3), its clock-sample and two-sets-of-sums instantiations, k_rows with
# X staged in LDS (BSR_NO_LDS=0) and 4 / 8 rows per lane (BSR_P1_U) -- which the byte-equality tests load through
# BSR_LIB_PATH (tests/test_gpu_stream.py, tests/test_gpu_edges.py).  `build.sh all` = both.
# Every translation unit is compiled on its own (in parallel), then linked; objects live in csrc/build/ (build_variants/).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ROCM="${ROCM_PATH:-/opt/rocm}"
what="${1:-ship}"
srcs=(bsr_tile bsr_tile_asm bsr_stream bsr_kernels bsr_api bsr_stage bsr_place bsr_comm bsr_engine bsr_refresh bsr_aql)

build_one() {   # <object dir> <output .so> [extra flags...]
  local obj="$1" out="$2"; shift 2
  mkdir -p "$obj"
  local FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -structurizecfg-skip-uniform-regions=true
               -ffunction-sections -fdata-sections -Wall -Wno-unused-function -Wno-inline-asm ${BSR_EXTRA_FLAGS:-} "$@")
  local pids=() s
  for s in "${srcs[@]}"; do
    [ -f "$here/$s.hip" ] || continue
    # the compiler's per-kernel resource report (registers, scratch) is kept next to the object: tests/test_build_resources.py
    # fails when the tile row pass spills to scratch (it did once, silently, for 4.5 us per launch)
    # (host code of the units that only launch kernels or run once per context: optimised for size)
    local hostopt=(); case "$s" in bsr_api|bsr_stage) ;; bsr_engine) hostopt=(-Xarch_host -O2) ;; *) hostopt=(-Xarch_host -Os) ;; esac
    # k_solve / k_rows / k_finalize: the first sixteen words of the kernel arguments arrive in scalar registers with the wave
    # (gfx940+ kernarg preload; older firmware runs the compiler's compatibility prologue, which asks for all of them at
    # once) -- these kernels read their arguments piece by piece, a scalar-load round trip each, at the head of a wave whose
    # whole life is a few microseconds
    case "$s" in bsr_kernels) hostopt+=(-mllvm -amdgpu-kernarg-preload-count=16) ;; esac
    "$ROCM/bin/hipcc" "${FLAGS[@]}" "${hostopt[@]}" -Rpass-analysis=kernel-resource-usage -c "$here/$s.hip" -o "$obj/$s.o" 2> "$obj/$s.resources.txt" &
    pids+=($!)
  done
  local rc=0 p
  for p in "${pids[@]}"; do wait "$p" || rc=1; done
  if [ $rc -ne 0 ]; then grep -h -B2 -A6 "error" "$obj"/*.resources.txt >&2 || true; exit 1; fi
  grep -h "warning:" "$obj"/*.resources.txt >&2 || true
  # scratch (private memory) instructions per function of the tile pass, counted in the disassembly: the resource report's
  # ScratchSize also counts stack slots the register allocator reserves and never touches
  local objdump="$ROCM/lib/llvm/bin/llvm-objdump"
  if [ -x "$objdump" ]; then
    local tmp; tmp="$(mktemp -d)"
    cp "$obj/bsr_tile.o" "$tmp/t.o"
    (cd "$tmp" && "$objdump" --offloading t.o > /dev/null && "$objdump" -d t.o.*gfx950 |
       awk '/^[0-9a-f]+ <.*>:$/ { name = $2; n[name] += 0 } /[ \t]scratch_(load|store)/ { n[name]++ } END { for (k in n) print k, n[k] }') \
      | sort > "$obj/bsr_tile.scratch_ops.txt" || true
    rm -rf "$tmp"
  fi
  local objs=()
  for s in "${srcs[@]}"; do [ -f "$obj/$s.o" ] && [ -f "$here/$s.hip" ] && objs+=("$obj/$s.o"); done
  # (the shipped library without its static symbol table: the C ABI's dynamic symbols stay)
  local strip=(); [ "$obj" = "$here/build" ] && strip=(-Wl,-s)
  "$ROCM/bin/hipcc" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -L"$ROCM/lib" -lrccl -lhsa-runtime64 -Wl,--gc-sections -Wl,--version-script="$here/exports.map" -Wl,-z,noseparate-code -Wl,-rpath,"$ROCM/lib" "${strip[@]}" -o "$out"
  echo "built $out ($(stat -c %s "$out") bytes)"
}

if [ "$what" = ship ] || [ "$what" = all ]; then build_one "$here/build" "$here/../bsr/libbsr_hip.so"; fi
if [ "$what" = variants ] || [ "$what" = all ]; then build_one "$here/build_variants" "$here/../bsr/libbsr_hip_variants.so" -DBSR_TEST_VARIANTS; fi
