#!/bin/bash
# TEST INFRASTRUCTURE.  CPU-only sanitizer builds of the product's host-side sampler (csrc/bsr_engine.hip, compiled as
# C++ with -DBSR_HOST_ONLY) against the CPU stand-in of the data side (stub_scorer.cpp).  No GPU, no HIP.
#   build_san.sh asan <out.so>   shared library, -fsanitize=address,undefined  (loaded by Python under LD_PRELOAD=libasan)
#   build_san.sh tsan <out>      executable around engine_tsan_main.cpp, -fsanitize=thread
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$here/../.."
mode="$1"; out="$2"
COMMON=(-std=c++17 -O1 -g -fno-omit-frame-pointer -DBSR_HOST_ONLY -pthread -I"$root/include")
if [ "$mode" = asan ]; then
  g++ "${COMMON[@]}" -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined \
      -x c++ "$root/mcmc-symreg_amd/csrc/bsr_engine.hip" -x c++ "$here/stub_scorer.cpp" -o "$out"
elif [ "$mode" = tsan ]; then
  g++ "${COMMON[@]}" -fsanitize=thread \
      -x c++ "$root/mcmc-symreg_amd/csrc/bsr_engine.hip" -x c++ "$here/stub_scorer.cpp" -x c++ "$here/engine_tsan_main.cpp" -o "$out"
else
  echo "usage: build_san.sh asan|tsan <out>" >&2; exit 2
fi
