#!/usr/bin/env python3
"""Where a k_solve wave spends its cycles (a library built with BSR_EXTRA_FLAGS=-DBSR_SOLVE_STAMPS: csrc/bsr_solve.h).
   usage: python3 tools/probes/solve_stamps.py [c2|c3]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
import numpy as np
import bench
wl_name = sys.argv[1] if len(sys.argv) > 1 else "c3"
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0, workload=wl_name, depth=1)
ranks = bench.Ranks()
wl = bench.build_workload(wl_name, args, ranks)
bench.generate_batches(wl, 4)
ctx = wl["ctx"]
names = ["loads", "wave sums", "to solve", "step 1 (QR)", "step 2 (bounds)", "step 3 (ridge)", "tail", "all"]
acc = []
for rep in range(3):
    for r in wl["packed"]:
        t = ctx.score_submit(r[0], r[1], r[2], r[3], r[4])
        ctx.score_wait(t, r[5])
        o = r[5]
        ok = (o["rank"] == wl["K"]) & ((o["flags"] & 16) != 0)
        if rep > 0:
            acc.append(o["beta"][ok][:, :8])
a = np.concatenate(acc)
print("%s: %d proposals settled by the fast tier; shader-clock cycles per wave (mean / median)" % (wl_name, len(a)))
for i, n in enumerate(names):
    print("  %-16s %9.0f %9.0f" % (n, a[:, i].mean(), np.median(a[:, i])))
wl["scorer"].close()
