cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
out=gpurun_out/tailprof_$v; rm -rf $out; mkdir -p $out
for depth in 1 6; do
BSR_FUSED_TAIL=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out/d$depth -- python3 bench.py --steps 200 --warmup 5 --cpu-sample 0 --extras 0 --depth $depth > $out/bench_d$depth.json 2> $out/err_d$depth.txt
python3 - $out/d$depth <<'PY'
import csv, glob, sys
st = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
for r in list(csv.DictReader(open(st[0])))[:6]:
    print("%-60s calls %6s avg %9.1f ns  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
done
done
