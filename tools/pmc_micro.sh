cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CASE="${CASE:-x1}"
rm -rf gpurun_out/pm1 gpurun_out/pm2 gpurun_out/pm3
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pm1 -- python3 tools/microbench_pass1.py > /dev/null 2>gpurun_out/pm1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pm2 -- python3 tools/microbench_pass1.py > /dev/null 2>gpurun_out/pm2.err
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --output-format csv -d gpurun_out/pm3 -- python3 tools/microbench_pass1.py > /dev/null 2>gpurun_out/pm3.err
python3 - <<'PY'
import csv, glob, collections
for d in ("pm1","pm2","pm3"):
    f = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not f: print(d, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:34]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key=(k,r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k]+=1
    for k, v in acc.items():
        if "k_rows" in k and n[k] > 5:
            print(d, k, "launches", n[k], {a: round(b / n[k]) for a, b in v.items()})
PY
