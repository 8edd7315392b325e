# Counters of the work-queue row pass (k_rows<..., PROJECT>) at a workload the tile pass does not take (default c5),
# with and without derived columns.  usage: bash tools/pmc_rows.sh [workload]
w=${1:-c5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dv in 0 1; do
  export BSR_DERIVED=$dv
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf gpurun_out/pa
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pa -- python3 bench.py --workload $w --steps 12 --warmup 3 --cpu-sample 0 --extras 0 --min-time 0 > /dev/null 2>/dev/null
    python3 - "$dv" <<'PY'
import csv, glob, collections, sys
f = glob.glob("gpurun_out/pa/*/*counter_collection.csv")
acc = collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if not ("k_rows" in k and ", 0>" in k and "<double, 0" not in k and "<float, 0" not in k): continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
n=max(1,len(seen))
print("derived %s launches %d " % (sys.argv[1], n), {k.replace("SQ_",""): round(v/n/1e3) for k,v in sorted(acc.items())}, "(thousands per launch; FETCH/WRITE_SIZE in KB/1000, fetch to be doubled on gfx950)")
PY
  done
done
rm -rf gpurun_out/pa
