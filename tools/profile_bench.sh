#!/bin/bash
# Run on the MI355X box: rocprofv3 kernel stats + HBM traffic counters (separate PMC passes) of the bench command.
# usage: bash tools/profile_bench.sh <tag> [bench args...]
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
# --depth 1: one batch at a time.  With several batches in flight the kernels of different streams share the GPU and
# the profiler's per-kernel duration includes the time a kernel waits for CUs (at N=1M two overlapping row passes each
# read 2x their isolated time); the roofline figure is the kernel on its own, as bench.py's isolated events time it.
args="--steps 50 --warmup 5 --cpu-sample 0 --depth 1 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $args > $out/bench_stats.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py $args > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py $args > /dev/null 2> $out/write.err
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
st = glob.glob(out + "/stats/*/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(out + "/kernel_stats.csv", "w") as f:
        f.write(open(st[0]).read())
    for r in rows[:8]:
        print("%-70s calls %5s avg %9.1f ns  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
res = {}
for name in ("fetch", "write"):
    f = glob.glob(out + "/%s/*/*counter_collection.csv" % name)
    if not f:
        continue
    acc = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        acc[k] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    for k in acc:
        # the projection pass: the tile kernel (bsr_tile.hip) or, where it does not apply, k_rows<..., PROJECT>
        if "k_tile" in k or "k_stream" in k or ("k_rows" in k and ", 0>" in k):
            res.setdefault(k, {})[name] = acc[k] / n[k]; res[k]["launches_" + name] = n[k]
summary = {}
for k, v in res.items():
    if v.get("launches_fetch", 0) < 10:
        continue
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
    # (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact
    fetch_b = v.get("fetch", 0.0) * 1024 * 2
    write_b = v.get("write", 0.0) * 1024
    summary[k] = {"fetch_bytes_per_launch_corrected": fetch_b, "write_bytes_per_launch": write_b,
                  "traffic_bytes_per_launch": fetch_b + write_b, "launches": v.get("launches_fetch")}
    print(k[:60], json.dumps(summary[k]))
json.dump(summary, open(out + "/traffic.json", "w"), indent=1)
PY
tail -c 1500 $out/bench_stats.json
