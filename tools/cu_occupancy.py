#!/usr/bin/env python3
"""CU occupancy of the tile row pass by WORKGROUP LIFETIME, un-traced, at the bench's pipelined regime (VERDICT r5 #1a).

    BSR_TILE_STAMPS=256 python tools/cu_occupancy.py [--workload c2] [--depth 8] [--steps 4000]

With BSR_TILE_STAMPS=R the library keeps the per-wave stamps of the last R tile launches: start / end on the 100 MHz
clock and the hardware ids of the CU the wave ran on (XCC, SE, SH, CU).  The batches are submitted exactly as bench.py
does (that many in flight, direct dispatch); no profiler is attached.  Over the window covered by complete launches the
tool prints, per CU: busy time (union of the workgroups' [first wave start, last wave end]) / window; the gap between one
workgroup's end and the next one's start on the same CU; workgroup lifetime; how many CUs ever hosted a tile workgroup.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
os.environ.setdefault("BSR_TILE_STAMPS", "256")

import numpy as np

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--chains", type=int, default=0)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4000)
    a = ap.parse_args()
    args = argparse.Namespace(batch=a.batch, chains=a.chains, dtype="f64", burnin=300)
    ranks = bench.Ranks()
    wl = bench.build_workload(a.workload, args, ranks)
    bench.generate_batches(wl, 8)
    ctx = wl["ctx"]
    tickets = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        r = wl["packed"][i % 8]
        tickets.append((ctx.score_submit_prepared(r[7]), r))
        if len(tickets) >= a.depth:
            t, rr = tickets.pop(0)
            ctx.score_wait_ptr(t, rr[8])
    while tickets:
        t, rr = tickets.pop(0)
        ctx.score_wait_ptr(t, rr[8])
    el = time.perf_counter() - t0
    print("%d steps, %d in flight: %.2f us per step (%.2f M proposals/s); dispatch %s" %
          (a.steps, a.depth, 1e6 * el / a.steps, wl["P"] * a.steps / el / 1e6, ctx.dispatch_info()))
    L = ctx._L
    L.bsr_debug_tile_stamp_ring.restype = C.c_int
    L.bsr_debug_tile_stamp_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    R = int(os.environ["BSR_TILE_STAMPS"])
    info = np.zeros(4, dtype=np.int32)
    probe = np.zeros((1, 1024, 16, 8), dtype=np.uint64)
    n = L.bsr_debug_tile_stamp_ring(ctx._h, probe.ctypes.data, 0, info.ctypes.data)
    R, slots, wgs, launches = [int(v) for v in info]
    if R <= 0:
        print("no stamps: BSR_TILE_STAMPS=R and the assembly tile pass (k_tile1a) are needed")
        return
    buf = np.zeros((R, slots, 16, 8), dtype=np.uint64)
    n = L.bsr_debug_tile_stamp_ring(ctx._h, buf.ctypes.data, R, info.ctypes.data)
    st = buf[:n, :wgs].astype(np.int64)
    t_s, t_e, hw = st[..., 7], st[..., 6], st[..., 5]
    ok = (t_s > 0) & (t_e >= t_s)
    print("ring of %d launches x %d workgroups x 16 waves; %d tile launches in all; waves with stamps: %d of %d" %
          (n, wgs, launches, ok.sum(), ok.size))
    # per workgroup: first wave start .. last wave end, on the CU of its waves
    big = np.iinfo(np.int64).max
    w_s = np.where(ok, t_s, big).min(axis=2)
    w_e = np.where(ok, t_e, 0).max(axis=2)
    has = ok.any(axis=2)
    cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)   # xcc, se, sh, cu
    wg_cu = np.where(ok, cu_key, -1).max(axis=2)
    same = np.array([[len(set(cu_key[l, w][ok[l, w]])) <= 1 for w in range(wgs)] for l in range(n)])
    print("workgroups whose 16 waves report one CU: %d of %d" % ((same & has).sum(), has.sum()))
    # the ring's oldest launches may be half overwritten: keep the launches whose every workgroup has stamps, and the window
    # in which ALL kept launches lie
    full = has.all(axis=1)
    l_s = np.where(has, w_s, big).min(axis=1)
    l_e = np.where(has, w_e, 0).max(axis=1)
    keep = np.where(full)[0]
    if len(keep) < 8:
        print("too few complete launches in the ring (%d)" % len(keep))
        return
    order = keep[np.argsort(l_s[keep])]
    # drop the first and last eighth (their neighbours in time are not in the ring: the CUs look idle around them)
    cut = max(1, len(order) // 8)
    inner = order[cut:-cut]
    win0, win1 = l_s[inner].min(), l_e[inner].max()
    life = ((w_e - w_s)[inner] * 0.01).reshape(-1)
    dur = (l_e - l_s)[inner] * 0.01
    print("launches kept: %d of %d; window %.1f us" % (len(inner), n, (win1 - win0) * 0.01))
    def stat(name, v):
        v = np.asarray(v, dtype=np.float64)
        print("%-44s min %7.2f  p10 %7.2f  median %7.2f  mean %7.2f  p90 %7.2f  max %7.2f us" %
              (name, v.min(), np.percentile(v, 10), np.median(v), v.mean(), np.percentile(v, 90), v.max()))
    stat("workgroup lifetime (first wave in .. last out)", life)
    stat("launch: first workgroup in .. last out", dur)
    # busy intervals per CU: all workgroups of ALL launches in the ring that overlap the window
    iv = {}
    for l in order:
        for w in range(wgs):
            if has[l, w]:
                iv.setdefault(int(wg_cu[l, w]), []).append((int(w_s[l, w]), int(w_e[l, w])))
    busy_tot, gaps, n_cu = 0.0, [], 0
    per_cu = []
    for cu, lst in iv.items():
        lst.sort()
        b = 0
        last_end = None
        cur_s, cur_e = None, None
        for s_, e_ in lst:
            s2, e2 = max(s_, win0), min(e_, win1)
            if last_end is not None and s_ >= win0 and e_ <= win1:
                gaps.append((s_ - last_end) * 0.01)
            last_end = e_ if last_end is None else max(last_end, e_)
            if e2 <= s2:
                continue
            if cur_e is None or s2 > cur_e:
                if cur_e is not None:
                    b += cur_e - cur_s
                cur_s, cur_e = s2, e2
            else:
                cur_e = max(cur_e, e2)
        if cur_e is not None:
            b += cur_e - cur_s
        per_cu.append(b / float(win1 - win0))
        busy_tot += b
        n_cu += 1
    per_cu = np.array(per_cu)
    print("CUs that hosted a tile workgroup: %d" % n_cu)
    print("CU occupancy by workgroup lifetime: %.3f of %d CUs x window (over 256 CUs: %.3f); per CU min %.2f median %.2f max %.2f" %
          (busy_tot / (n_cu * float(win1 - win0)), n_cu, busy_tot / (256 * float(win1 - win0)), per_cu.min(), np.median(per_cu), per_cu.max()))
    stat("gap on a CU: workgroup out .. next one in", gaps)
    g = np.array(gaps)
    print("  share of gaps < 0 (two tile workgroups on one CU at once): %.3f; < 1 us: %.3f; > 5 us: %.3f" %
          ((g < 0).mean(), (g < 1).mean(), (g > 5).mean()))
    print("tile workgroups per step x mean lifetime / (256 CUs x step): %.3f" % (wgs * life.mean() / (256 * 1e6 * el / a.steps)))
    wl["scorer"].close()


if __name__ == "__main__":
    main()
