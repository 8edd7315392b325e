#!/usr/bin/env python3
"""Per-wave clock stamps of the tile row pass (bsr_tile.hip) on the bench workload.  Run on the GPU box:

    BSR_TILE_STAMPS=1 python tools/tile_stamps.py [--workload c2] [--batch 64] [--chains 1]

Prints, over all waves of the last launch: when the workgroups started (spread), the time to stage the first chunk,
the compute time, the reduction/store time and the whole wave lifetime, in microseconds (shader clock from the ratio
of s_memtime to the 100 MHz s_memrealtime is not needed: stamps are converted with the clock measured over the
kernel)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
os.environ.setdefault("BSR_TILE_STAMPS", "1")

import numpy as np

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--chains", type=int, default=0)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--pipelined", type=int, default=0,
                    help="that many batches in flight for 3000 steps in front of the read-out: the stamps are then of "
                         "workgroups of the LAST launches, which ran beside other batches' kernels (clock, staging and "
                         "lifetime under load; the launch-wide figures -- start spread, end -- mix launches)")
    a = ap.parse_args()
    args = argparse.Namespace(batch=a.batch, chains=a.chains, dtype="f64", burnin=300)
    ranks = bench.Ranks()
    wl = bench.build_workload(a.workload, args, ranks)
    bench.generate_batches(wl, 8)
    ctx = wl["ctx"]
    for i in range(a.launches):
        r = wl["packed"][i % 8]
        ctx.set_profiling(1)
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], r[5])
        kern = ctx.last_timing()[0]
    if a.pipelined > 0:
        tickets = []
        for i in range(3000):
            r = wl["packed"][i % 8]
            tickets.append((ctx.score_submit_prepared(r[7]), r))
            if len(tickets) >= a.pipelined:
                t, rr = tickets.pop(0)
                ctx.score_wait_ptr(t, rr[8])
        print("pipelined: %d batches in flight AT the read-out" % a.pipelined)
    L = ctx._L
    L.bsr_debug_tile_stamps.restype = C.c_int
    L.bsr_debug_tile_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    buf = np.zeros((1024, 16, 8), dtype=np.uint64)
    geom = np.zeros(5, dtype=np.int32)
    n = L.bsr_debug_tile_stamps(ctx._h, buf.ctypes.data, 1024, geom.ctypes.data)
    if a.pipelined > 0:
        while tickets:
            t, rr = tickets.pop(0)
            ctx.score_wait_ptr(t, rr[8])
    if n <= 0:
        print("no stamps (BSR_TILE_STAMPS=1 and a tile launch are needed)", n)
        return
    raw = buf[:n]
    T, n_slices, bps, n_blocks, n_cu = [int(v) for v in geom]
    n_cu, n_sub = n_cu // 100, n_cu % 100
    print("geometry: T=%d slices=%d blocks/slice=%d sub-slices=%d blocks=%d CUs=%d; kernel %.2f us by HIP events" %
          (T, n_slices, bps, n_sub, n_blocks, n_cu, kern))
    st = raw.astype(np.int64)
    # waves that worked in this launch: all five shader-clock stamps present and increasing
    ok = np.ones(st.shape[:2], dtype=bool)
    for i in range(4):
        ok &= st[:, :, i + 1] >= st[:, :, i]
    ok &= st[:, :, 4] > st[:, :, 0]
    ok &= st[:, :, 0] > 0
    print("waves with work: %d of %d (%d workgroups)" % (ok.sum(), ok.size, ok.any(axis=1).sum()))
    rt0 = st[:, :, 7][ok].astype(np.float64) * 0.01      # us
    rt1 = st[:, :, 6][ok].astype(np.float64) * 0.01
    life_cyc = (st[:, :, 4] - st[:, :, 0])[ok].astype(np.float64)
    mhz = np.median(life_cyc / np.maximum(rt1 - rt0, 0.05))
    print("shader clock %.0f MHz (median over waves of cycles / 100 MHz-clock time)" % mhz)

    def stat(name, v):
        v = np.asarray(v, dtype=np.float64)
        print("%-36s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us" %
              (name, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
    base = rt0.min()
    stat("wave start (100 MHz clock)", rt0 - base)
    stat("wave end (100 MHz clock)", rt1 - base)
    d = lambda a, b: (st[:, :, b] - st[:, :, a])[ok] / mhz
    stat("stage first chunk (0->1)", d(0, 1))
    stat("compute first chunk (1->2)", d(1, 2))
    stat("all chunks incl. later staging (1->3)", d(1, 3))
    stat("reduce + store (3->4)", d(3, 4))
    if (st[:, :, 5][ok] > 0).any():   # chunked kernel: cycles a wave ran tapes, the rest of 1->3 it waited (copies, barriers)
        busy = st[:, :, 5].astype(np.float64) / mhz
        stat("running tapes (sum over chunks)", busy[ok])
        print("  running tapes by wave index (mean over workgroups, us): " +
              " ".join("%.0f" % np.mean([busy[w][i] for w in range(n) if ok[w][i]]) for i in range(busy.shape[1]) if ok[:, i].any()))
        per = np.array([busy[w][ok[w]].max() / max(1e-9, busy[w][ok[w]].mean()) for w in range(n) if ok[w].any()])
        print("busy imbalance inside a workgroup (busiest wave / mean wave): median %.2f  max %.2f" % (np.median(per), per.max()))
    stat("wave lifetime (0->4)", d(0, 4))
    comp = (st[:, :, 3] - st[:, :, 1]).astype(np.float64)
    per_wg = np.array([comp[w][ok[w]].max() / max(1e-9, comp[w][ok[w]].mean()) for w in range(n) if ok[w].any()])
    print("compute imbalance inside a workgroup (slowest wave / mean wave): median %.2f  max %.2f" %
          (np.median(per_wg), per_wg.max()))
    wgs = [w for w in range(n) if ok[w].any()]
    wg_mean = np.array([comp[w][ok[w]].mean() for w in wgs]) / mhz
    wg_max = np.array([comp[w][ok[w]].max() for w in wgs]) / mhz
    stat("per workgroup: mean wave compute", wg_mean)
    stat("per workgroup: slowest wave compute", wg_max)
    for g in range(T):
        sel = np.array([w // n_slices == g for w in wgs])
        if sel.any():
            print("  tape group %d: workgroup mean compute median %.2f us, slowest wave median %.2f us, max %.2f us" %
                  (g, np.median(wg_mean[sel]), np.median(wg_max[sel]), wg_max[sel].max()))
    slow = np.argsort(-wg_max)[:8]
    print("  slowest workgroups (id, xcd=id%%8, slice, group): %s" %
          ", ".join("(%d,%d,%d,%d: %.1f us)" % (wgs[i], wgs[i] % 8, wgs[i] % n_slices, wgs[i] // n_slices, wg_max[i]) for i in slow))
    wl["scorer"].close()


if __name__ == "__main__":
    main()
