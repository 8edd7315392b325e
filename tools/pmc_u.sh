cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for u in 2 4; do
export BSR_P1_U=$u
rm -rf gpurun_out/pu$u
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pu$u -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > /dev/null 2>gpurun_out/pu$u.err
python3 - $u <<'PY'
import csv, glob, collections, sys
u = sys.argv[1]
f = glob.glob("gpurun_out/pu%s/*/*counter_collection.csv" % u)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "k_rows" not in k or ", 0>" not in k: continue
    k = k[:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
for k, v in acc.items():
    if n[k] > 5:
        d = {a: round(b / n[k]) for a, b in v.items()}
        tot = d['SQ_INSTS_VALU'] + d['SQ_INSTS_SALU'] + d['SQ_INSTS_BRANCH'] + d['SQ_INSTS_SMEM'] + d['SQ_INSTS_VMEM_RD']
        print("U", u, k, n[k], d, "total", tot, "predicted us at 1 instr / 4 cycles / SIMD @2.1GHz: %.1f" % (tot * 4 / 1024 / 2100.0))
PY
done
