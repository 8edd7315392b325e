#!/bin/bash
# What the native sampler (8 chains on one GPU: config 4's per-GPU share) consumes per second with 2, 4, 6, 8 and all CPUs,
# with and without the groups' helper threads: the prediction for a rank of the 8-GPU run.
#   bash tools/engine_cpus.sh [out.txt]
out=${1:-/dev/stdout}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
# CPUs of the GPU's NUMA node first (the library pins its threads there: csrc/bsr_place.h)
node=$(python3 - <<'PY'
import sys, os
sys.path.insert(0, "mcmc-symreg_amd")
from bsr import _lib
import numpy as np
pl = np.zeros(4, dtype=np.int32)
_lib.lib().bsr_place_info(_lib.ptr(pl))
print(int(pl[2]))
PY
)
cpus=$(cat /sys/devices/system/node/node${node}/cpulist 2>/dev/null || echo "0-$(($(nproc)-1))")
first=$(python3 -c "
s='$cpus'; out=[]
for p in s.split(','):
    a,_,b=p.partition('-'); out+=list(range(int(a), int(b or a)+1))
import os
ok=sorted(os.sched_getaffinity(0)); out=[c for c in out if c in ok] or ok
print(','.join(str(c) for c in out[:32]))")
echo "GPU on NUMA node $node; CPUs used for the masks: $first (quota: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null))" | tee -a $out
run() {   # label, n_cpus (0 = no mask), env...
  local label=$1 n=$2; shift 2
  local mask=""
  if [ "$n" != 0 ]; then mask="taskset -c $(echo $first | cut -d, -f1-$n)"; fi
  line=$(env "$@" $mask python3 - <<'PY' 2>&1 | tail -1
import sys, os, argparse
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
b = bench.engine_leg(args, ranks)
a = bench.engine_leg(args, ranks, chains=1, seconds=2.0)
print("eight chains %.2f M consumed/s (memo %.2f, discarded %.3f, %s threads); one chain %.2f M" % (
    b["value"] / 1e6, b.get("memo_answered_fraction_of_generated", 0), b["discarded_fraction"], b.get("threads", "?"), a["value"] / 1e6))
PY
)
  echo "$label | cpus $n | $* | $line" | tee -a $out
}
for n in 0 12 10 8 6 4 2; do
  run "default (four groups, helper threads as the CPUs allow)" $n X=1
done
run "no helper threads" 0 BSR_ENGINE_HELPERS=0
run "no helper threads" 8 BSR_ENGINE_HELPERS=0
run "groups 8" 0 BSR_ENGINE_GROUPS=8
