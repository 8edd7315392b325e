#!/usr/bin/env python3
"""Per-kernel times of a batch (HIP events around every kernel, one batch at a time) and the flagged-proposal count.
Run on the GPU box:  python tools/probes/tail_times.py [--workload c2]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
import numpy as np
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="c2"); a = ap.parse_args()
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
wl = bench.build_workload(a.workload, args, ranks)
bench.generate_batches(wl, 32)
ctx = wl["ctx"]
acc = np.zeros(5); n = 0; flagged = []
for rep in range(4):
    for r in wl["packed"]:
        ctx.set_profiling(2)
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], r[5])
        t = ctx.last_timing()
        if rep > 0:
            acc += t; n += 1
            flagged.append(int(np.sum((r[5]["flags"] & 0x40) != 0)) if "flags" in r[5].dtype.names else 0)
ctx.set_profiling(0)
acc /= n
print("%s: row %.1f  solve %.1f  residual %.1f  finalize(+events) %.1f  total %.1f us" % (a.workload, *acc))
r = wl["packed"][0][5]
print("rank<K share %.3f" % float(np.mean(r["rank"] < wl["K"])))
