# PMC passes over the bench command (run on the GPU box): instruction mix, issue/wait cycles and LDS behaviour of the
# tile row pass (k_tile / k_tile1).  usage: bash tools/pmc_tile.sh [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--steps 20 --warmup 3 --cpu-sample 0 --extras 0 --min-time 0 $@"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_IFETCH SQ_INST_LEVEL_LDS"; do
  i=$((i+1)); rm -rf gpurun_out/pt$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pt$i -- python3 bench.py $args > /dev/null 2>gpurun_out/pt$i.err || tail -3 gpurun_out/pt$i.err
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for i in range(1, 5):
    f = glob.glob("gpurun_out/pt%d/*/*counter_collection.csv" % i)
    if not f: print(i, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "k_tile" not in k and "k_stream" not in k: continue
        import re as _re; k = (_re.search(r"k_(tile1a|tile1|tile|stream)<[^>]*>", k) or _re.search(r".{0,60}$", k)).group(0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    for k, v in acc.items():
        if n[k] > 5:
            d = {a: round(b / n[k]) for a, b in v.items()}
            out.setdefault(k, {"launches": n[k]}).update(d)
            print(i, k, "launches", n[k], d)
json.dump(out, open("gpurun_out/pmc_tile.json", "w"), indent=1)
PY
