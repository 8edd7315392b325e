# Kernel timeline of the pipelined bench region (run on the GPU box): rocprofv3 --kernel-trace, then per-kernel
# durations, the GPU-busy fraction and how much of each step the row pass runs alone.
# usage: bash tools/timeline.sh [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 400 --warmup 20 --cpu-sample 0 --extras 0 --min-time 0 "$@" > gpurun_out/tl_bench.json 2> gpurun_out/tl.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tl/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = "tile" if ("k_tile" in n or "k_stream" in n) else "solve" if n.startswith("k_solve") else "finalize" if n.startswith("k_finalize") else \
            "residual" if ("k_rows" in n and ", 1>" in n) else "rows" if "k_rows" in n else "events" if "k_events" in n else "other"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
ev.sort()
# the steady part: last 60 % of the tile launches
tiles = [e for e in ev if e[2] == "tile"]
t0 = tiles[int(len(tiles) * 0.4)][0]; t1 = tiles[-1][1]
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
n_tile = sum(1 for e in sel if e[2] == "tile")
span = (t1 - t0) / 1e3
print("steady window %.1f us, %d row passes -> %.2f us per step" % (span, n_tile, span / n_tile))
dur = collections.defaultdict(list)
for s, e, k, q in sel: dur[k].append((e - s) / 1e3)
for k, v in sorted(dur.items()):
    print("  %-9s launches %5d  mean %.2f us  sum/step %.2f us" % (k, len(v), sum(v) / len(v), sum(v) / n_tile))
# busy time: union of intervals; time with only small kernels running; time with nothing running
pts = []
for s, e, k, q in sel:
    pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort()
active = collections.Counter(); last = t0; idle = only_small = tile_alone = tile_multi = tile_plus_small = 0
for t, d, k in pts:
    dt = t - last
    if dt > 0:
        nt = active["tile"]; ns = sum(v for kk, v in active.items() if kk != "tile")
        if nt == 0 and ns == 0: idle += dt
        elif nt == 0: only_small += dt
        elif nt == 1 and ns == 0: tile_alone += dt
        elif nt >= 2: tile_multi += dt
        else: tile_plus_small += dt
    active[k] += d; last = t
tot = (t1 - t0)
print("share of the window: idle %.1f%%, only small kernels %.1f%%, one row pass alone %.1f%%, row pass + small kernels %.1f%%, two or more row passes %.1f%%"
      % (100 * idle / tot, 100 * only_small / tot, 100 * tile_alone / tot, 100 * tile_plus_small / tot, 100 * tile_multi / tot))
# how much of the machine the row passes hold: their workgroups take a whole CU each (grid size from the trace)
wg = {}
for r in rows:
    n = r["Kernel_Name"]
    if "k_tile" in n or "k_stream" in n:
        try:
            wg[int(r["Start_Timestamp"])] = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        except Exception:
            pass
cu_us = sum(((e - s) / 1e3) * wg.get(s, 97) for s, e, k, q in sel if k == "tile")
print("row passes: %.0f CU-us per batch (launch duration x workgroups), %.1f%% of 256 CUs x window" % (cu_us / n_tile, 100 * cu_us / (256 * span)))
# row passes in flight, launch by launch: start, end, how many other row passes overlap it
tl = [(s, e) for s, e, k, q in sel if k == "tile"]
print("consecutive row passes (us from window start): start -> end, others in flight at its start / overlapping it at all")
for i, (s, e) in enumerate(tl[:220]):
    at_start = sum(1 for s2, e2 in tl if s2 < s < e2)
    overl = sum(1 for s2, e2 in tl if (s2, e2) != (s, e) and s2 < e and e2 > s)
    print("   %4d  %9.2f -> %9.2f   %d / %d" % (i, (s - t0) / 1e3, (e - t0) / 1e3, at_start, overl))
print("sample of all kernels (us from window start):")
for s, e, k, q in sel[:24]:
    print("   %8.2f -> %8.2f  %-9s queue %s" % ((s - t0) / 1e3, (e - t0) / 1e3, k, q))
PY
rm -rf gpurun_out/tl
