mkdir -p gpurun_out/r05g
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows 2048 --min-time 1 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("rows2048 $label", round(d["value"]), round(d["ms_per_step"]*1000,2))
except Exception as e:
    print("$label failed", e); print(open("gpurun_out/r05g/x.err").read()[-600:])
PY
}
run "hip" BSR_AQL=0
run "aql fence2 q4" BSR_AQL_FENCE=2
run "aql fence2 q8" BSR_AQL_FENCE=2 BSR_AQL_QUEUES=8
run "aql fence2 q6" BSR_AQL_FENCE=2 BSR_AQL_QUEUES=6
run "aql fence2 q4 sig0" BSR_AQL_FENCE=2 BSR_AQL_SIGNAL=0
run "aql fence1 q4" BSR_AQL_FENCE=1
timeout 900 python -m pytest tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | grep "passed\|failed" | tail -2
# gaps between the kernels of a batch, both ways
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in 1 0; do
rm -rf gpurun_out/tlq
BSR_AQL=$mode rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlq -- python3 bench.py --steps 300 --warmup 20 --cpu-sample 0 --extras 0 --min-time 0 --rows 2048 > /dev/null 2> gpurun_out/r05g/tl_$mode.err
python3 - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/tlq/*/*kernel_trace.csv")
if not fs:
    print("no trace for mode $mode"); raise SystemExit
rows = list(csv.DictReader(open(fs[0])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = "tile" if ("k_tile" in n or "k_stream" in n or ("k_rows" in n and ", 0>" in n)) else "solve" if n.startswith("k_solve") else "residual" if ("k_rows" in n and ", 1>" in n) else "other"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
ev.sort()
ev = ev[len(ev)//2:]
byq = collections.defaultdict(list)
for e in ev: byq[e[3]].append(e)
g1=[];g2=[];d=collections.defaultdict(list)
for q, L in byq.items():
    for a, b in zip(L, L[1:]):
        if a[2]=="tile" and b[2]=="solve": g1.append((b[0]-a[1])/1e3)
        if a[2]=="solve" and b[2]=="residual": g2.append((b[0]-a[1])/1e3)
for e in ev: d[e[2]].append((e[1]-e[0])/1e3)
import statistics as st
print("AQL=$mode queues", len(byq), "tile->solve gap median %.2f us, solve->residual gap median %.2f us" % (st.median(g1) if g1 else -1, st.median(g2) if g2 else -1),
      {k: round(st.median(v),2) for k,v in d.items()})
PY
done
rm -rf gpurun_out/tlq
