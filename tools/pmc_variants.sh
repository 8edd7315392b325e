# A/B of library builds on one box: instruction counters of the tile row pass (one PMC pass) and the bench's own
# event timing, per variant.  usage: bash tools/pmc_variants.sh <name>...   (mcmc-symreg_amd/bsr/libbsr_<name>.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_$v.so
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
print("variant %-10s launches %d " % (sys.argv[1], n), {k.replace("SQ_INSTS_",""): round(v/n/1e3) for k,v in sorted(acc.items())}, "(thousands per launch)")
PY
done
rm -rf gpurun_out/pa
for i in 1 2 3; do
for v in "$@"; do
  BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_$v.so python bench.py --steps 200 --warmup 20 --extras 0 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-10s'%'$v', round(d['value']), 'row pass %.2f us isolated, %.2f us in the pipelined region'%(d['roofline']['kernel_us'], d['roofline']['kernel_us_in_timed_region']))"
done; done
