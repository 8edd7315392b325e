#!/usr/bin/env python3
"""How much of the headline is the HOST's: N caller threads, each with its own context on the one GPU (same data, same
frozen chain state), each running bench.py's pipelined step loop; ctypes releases the GIL inside the C calls.  Prints
the aggregate proposals/s for 1, 2 and 3 callers.  Run on the GPU box: python tools/probes/two_callers.py [--rows N]"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--callers", type=int, default=3)
    a = ap.parse_args()
    args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=a.rows)
    ranks = bench.Ranks()
    wls = []
    for i in range(a.callers):
        wl = bench.build_workload("c2", args, ranks)
        bench.generate_batches(wl, 64)
        wls.append(wl)

    def loop(wl, stop, out, k):
        ctx, packed = wl["ctx"], wl["packed"]
        n_unique = len(packed)
        tickets = []
        n = 0
        while not stop[0]:
            r = packed[n % n_unique]
            tickets.append((ctx.score_submit_prepared(r[7]), r))
            if len(tickets) >= a.depth:
                t, rr = tickets.pop(0)
                ctx.score_wait_ptr(t, rr[8])
            n += 1
        while tickets:
            t, rr = tickets.pop(0)
            ctx.score_wait_ptr(t, rr[8])
        out[k] = n

    for n_callers in range(1, a.callers + 1):
        for phase in ("warm", "timed"):
            stop, out = [False], [0] * n_callers
            ths = [threading.Thread(target=loop, args=(wls[k], stop, out, k)) for k in range(n_callers)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            time.sleep(0.4 if phase == "warm" else a.seconds)
            stop[0] = True
            for th in ths:
                th.join()
            dt = time.perf_counter() - t0
        B = wls[0]["B"]
        print("%d caller(s): %.2f M proposals/s aggregate (%.2f us per step and caller; steps %r)"
              % (n_callers, sum(out) * B / dt / 1e6, dt / max(1, max(out)) * 1e6, out), flush=True)


if __name__ == "__main__":
    main()
