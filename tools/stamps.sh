# In-kernel clock stamps of the row pass (run on the GPU box): builds a -DBSR_STAMPS copy of the library in the scratch
# checkout, swaps it in for this command only, and prints the per-phase cycle budget of a wave (tools/stamps.py).
set -e
cd "$GRAFT_REPO_ROOT"
cs=mcmc-symreg_amd/csrc
cp mcmc-symreg_amd/bsr/libbsr_hip.so /tmp/libbsr_hip.orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -structurizecfg-skip-uniform-regions=true \
  -DBSR_STAMPS -Wno-unused-function $cs/bsr_kernels.hip $cs/bsr_api.hip $cs/bsr_engine.hip $cs/bsr_refresh.hip \
  -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o mcmc-symreg_amd/bsr/libbsr_hip.so
for c in "" x1 neg4 sin 4term; do CASE="$c" python tools/stamps.py 2>&1 | grep -E "case|kernel us|wall clock|waves with|histogram|lifetime|setup|sweep|reductions"; done
cp /tmp/libbsr_hip.orig.so mcmc-symreg_amd/bsr/libbsr_hip.so
