cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BSR_NO_LDS=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > /dev/null 2>gpurun_out/pmc1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > /dev/null 2>gpurun_out/pmc2.err
tail -3 gpurun_out/pmc1.err gpurun_out/pmc2.err
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1","pmc2"):
    f = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not f: print(d, "no file", glob.glob("gpurun_out/%s/*/*" % d)); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "pass1" in k or "solve" in k:
            print(d, k, {a: round(b) for a, b in v.items()})
PY
