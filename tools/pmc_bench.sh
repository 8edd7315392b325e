# PMC passes over the bench command (run on the GPU box): instruction mix, issue/wait cycles, L1/L2 behaviour of k_rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--steps 20 --warmup 3 --cpu-sample 0 ${BENCH_ARGS:-}"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MFMA_F64" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1)); rm -rf gpurun_out/pb$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pb$i -- python3 bench.py $args > /dev/null 2>gpurun_out/pb$i.err || tail -3 gpurun_out/pb$i.err
done
python3 - <<'PY'
import csv, glob, collections
for i in range(1, 7):
    f = glob.glob("gpurun_out/pb%d/*/*counter_collection.csv" % i)
    if not f: print(i, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "k_rows" not in k or not k.rstrip().endswith("0>(double const*, double const*, long, long, unsigned long const*, unsigned long const*, double const*, PropDesc const*, PropCoef const*, int, int, int, int, int, int const*, int, double*, double*, int)") and ", 0>" not in k: continue
        if ", 0>" not in k: continue
        k = k[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    for k, v in acc.items():
        if n[k] > 5: print(i, k, "launches", n[k], {a: round(b / n[k]) for a, b in v.items()})
PY
