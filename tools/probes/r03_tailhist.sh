cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
out=/tmp/tailprof_$v; rm -rf $out; mkdir -p $out
for depth in 1 6; do
BSR_FUSED_TAIL=$v rocprofv3 --kernel-trace --output-format csv -d $out/d$depth -- python3 bench.py --steps 200 --warmup 5 --cpu-sample 0 --extras 0 --depth $depth > $out/bench_d$depth.json 2> $out/err_d$depth.txt
python3 - $out/d$depth $v $depth <<'PY'
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    dur[r["Kernel_Name"][:34]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
for k, x in dur.items():
    if len(x) < 100: continue
    x = np.array(x)
    print("fused" if sys.argv[2]=="1" else "legacy", "depth", sys.argv[3], k, "n", len(x), "mean %.1f" % x.mean(), "pct 5/25/50/75/95:", np.percentile(x,[5,25,50,75,95]).round(1))
PY
done
done
