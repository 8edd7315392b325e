cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" REDUCE ACC TAPE; do
  lib=""; [ -n "$v" ] && lib="$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_ablate_$v.so"
  export BSR_LIB_PATH=$lib
  rm -rf gpurun_out/pa
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d gpurun_out/pa -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --extras 0 --min-time 0 > /dev/null 2>/dev/null
  python3 - "$v" <<'PY'
import csv, glob, collections, sys
f = glob.glob("gpurun_out/pa/*/*counter_collection.csv")
acc = collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f[0])):
    if "k_tile" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
n=len(seen)
print("variant %-7s launches %d " % (sys.argv[1] or "full", n), {k: round(v/n/1e3) for k,v in sorted(acc.items())}, "(thousands per launch)")
PY
done
rm -rf gpurun_out/pa
