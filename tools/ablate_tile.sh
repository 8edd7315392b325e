# Timing experiments on the tile row pass: rebuilds the library with parts of the kernel removed (results wrong,
# timing only) and prints the kernel time of the bench workload for each.  Build here, run on the GPU box:
#   bash tools/ablate_tile.sh build        (in the build container)
#   bash tools/ablate_tile.sh run          (on the GPU box)
here="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" = "build" ]; then
  for v in REDUCE ACC TAPE; do
    ( cd $here/mcmc-symreg_amd/csrc && mkdir -p build_$v && for s in bsr_tile bsr_kernels bsr_api bsr_engine bsr_refresh; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -structurizecfg-skip-uniform-regions=true -DBSR_ABLATE_$v -c $s.hip -o build_$v/$s.o & done; wait
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_$v/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../bsr/libbsr_ablate_$v.so ) 2>&1 | grep -v warning | grep -i error
  done
  ls -la $here/mcmc-symreg_amd/bsr/*.so
else
  cd $here
  for v in "" REDUCE ACC TAPE; do
    lib=""; [ -n "$v" ] && lib="$here/mcmc-symreg_amd/bsr/libbsr_ablate_$v.so"
    echo "== variant ${v:-full}"; BSR_LIB_PATH=$lib BSR_TILE_STAMPS=1 timeout 300 python tools/tile_stamps.py "${@:2}" 2>&1 | grep -E "geometry|wave end|all chunks|stage first"
  done
fi
