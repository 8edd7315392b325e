# Executed instruction counts per tape shape of the scoring row pass (run on the GPU box):
#   bash tools/probes/op_costs.sh [--N 1000000 --d 50] > gpurun_out/op_costs.txt
# Two counter passes over tools/probes/op_costs.py; the k-th group of `reps` row-pass dispatches belongs to shape k.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
reps=12
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf gpurun_out/oc$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/oc$i -- python3 tools/probes/op_costs.py --reps $reps "$@" > gpurun_out/oc$i.log 2>gpurun_out/oc$i.err || tail -3 gpurun_out/oc$i.err
done
python3 - "$reps" <<'PY'
import csv, glob, collections, re, sys
reps = int(sys.argv[1])
names = [l.split()[2] for l in open("gpurun_out/oc1.log") if l.startswith("shape ")]
times = [l.strip() for l in open("gpurun_out/oc1.log") if l.startswith("shape ")]
print(open("gpurun_out/oc1.log").readline().strip())
tab = collections.defaultdict(dict)
for i in (1, 2):
    f = glob.glob("gpurun_out/oc%d/*/*counter_collection.csv" % i)
    if not f:
        print(i, "no counter file"); continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if not re.search(r"k_tile|k_rows<.*0>|k_stream", r["Kernel_Name"]):
            continue
        if "k_rows" in r["Kernel_Name"] and ", 0>" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per.setdefault(d, {"kernel": r["Kernel_Name"][:50]})
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)
    # the scoring dispatches are the LAST len(names) * reps row-pass dispatches (context creation may run the pass too)
    ids = ids[-len(names) * reps:]
    for k, nm in enumerate(names):
        grp = ids[k * reps + 2:(k + 1) * reps]
        for c in per[grp[0]]:
            if c == "kernel": tab[nm]["kernel"] = per[grp[0]][c]; continue
            tab[nm][c] = sum(per[g][c] for g in grp) / len(grp)
for t in times: print(t)
cols = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT",
        "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"]
print("%-20s" % "shape", " ".join("%12s" % c.replace("SQ_INSTS_", "").replace("SQ_", "")[:12] for c in cols))
for nm in names:
    print("%-20s" % nm, " ".join("%12d" % round(tab[nm].get(c, -1)) for c in cols), tab[nm].get("kernel", ""))
PY
