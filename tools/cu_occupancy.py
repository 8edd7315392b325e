#!/usr/bin/env python3
"""CU occupancy of the tile row pass by WORKGROUP LIFETIME, un-traced, at the bench's pipelined regime (VERDICT r5 #1a).

    python tools/cu_occupancy.py [--workload c2] [--depth 8] [--steps 4000]      (BSR_TILE_STAMPS=R: ring size, default 256)

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

import numpy as np

import bench


def read_ring(ctx):
    """-> (stamps [launches][workgroups][16][8] int64, launches so far) of the context's stamp ring, or (None, 0)."""
    L = ctx._L
    L.bsr_debug_tile_stamp_ring.restype = C.c_int
    L.bsr_debug_tile_stamp_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    info = np.zeros(4, dtype=np.int32)
    probe = np.zeros(8, dtype=np.uint64)
    L.bsr_debug_tile_stamp_ring(ctx._h, probe.ctypes.data, 0, info.ctypes.data)
    R, slots, wgs, launches = [int(v) for v in info]
    if R <= 0 or wgs <= 0:
        return None, 0
    buf = np.zeros((R, slots, 16, 8), dtype=np.uint64)
    n = L.bsr_debug_tile_stamp_ring(ctx._h, buf.ctypes.data, R, info.ctypes.data)
    return buf[:n, :wgs].astype(np.int64), int(info[3])


def analyze(st, step_us=None):
    """Busy intervals per CU from the ring's stamps -> dict (times in us).  Word 7 / 6: a wave's start / end on the 100 MHz
    clock, word 5: HW_ID | XCC_ID << 32."""
    n, wgs = st.shape[:2]
    t_s, t_e, hw = st[..., 7], st[..., 6], st[..., 5]
    ok = (t_s > 0) & (t_e >= t_s)
    big = np.iinfo(np.int64).max
    w_s = np.where(ok, t_s, big).min(axis=2)
    w_e = np.where(ok, t_e, 0).max(axis=2)
    has = ok.any(axis=2)
    cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)   # xcc, se, sh, cu
    wg_cu = np.where(ok, cu_key, -1).max(axis=2)
    one_cu = int(sum(len(set(cu_key[l, w][ok[l, w]])) <= 1 for l in range(n) for w in range(wgs) if has[l, w]))
    # the ring's oldest launches may be half overwritten: keep the launches whose every workgroup has stamps; drop the
    # first and last eighth in time (their neighbours are not in the ring: the CUs look idle around them)
    full = has.all(axis=1)
    l_s = np.where(has, w_s, big).min(axis=1)
    l_e = np.where(has, w_e, 0).max(axis=1)
    keep = np.where(full)[0]
    if len(keep) < 8:
        return {"error": "too few complete launches in the ring (%d)" % len(keep)}
    order = keep[np.argsort(l_s[keep])]
    cut = max(1, len(order) // 8)
    inner = order[cut:-cut]
    win0, win1 = int(l_s[inner].min()), int(l_e[inner].max())
    life = ((w_e - w_s)[inner] * 0.01).reshape(-1)
    dur = (l_e - l_s)[inner] * 0.01
    iv = {}
    for l in order:
        for w in range(wgs):
            if has[l, w]:
                iv.setdefault(int(wg_cu[l, w]), []).append((int(w_s[l, w]), int(w_e[l, w])))
    busy_tot, gaps, per_cu = 0.0, [], []
    for cu, lst in iv.items():
        lst.sort()
        b, last_end, cur_s, cur_e = 0, None, None, None
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
    g = np.array(gaps) if gaps else np.zeros(1)
    per_cu = np.array(per_cu)
    q = lambda v, p: float(np.percentile(v, p))
    out = {"launches_in_ring": int(n), "launches_kept": int(len(inner)), "workgroups_per_launch": int(wgs),
           "window_us": (win1 - win0) * 0.01, "cus_hosting_tile_workgroups": len(iv),
           "workgroups_on_one_cu": one_cu, "workgroups_stamped": int(has.sum()),
           "occupancy_of_256_cus": busy_tot / (256.0 * (win1 - win0)),
           "occupancy_per_cu_min_median_max": [float(per_cu.min()), float(np.median(per_cu)), float(per_cu.max())],
           "workgroup_lifetime_us": {"p10": q(life, 10), "median": q(life, 50), "mean": float(life.mean()), "p90": q(life, 90)},
           "launch_first_in_to_last_out_us": {"p10": q(dur, 10), "median": q(dur, 50), "mean": float(dur.mean()), "p90": q(dur, 90)},
           "gap_on_a_cu_us": {"min": float(g.min()), "p10": q(g, 10), "median": q(g, 50), "mean": float(g.mean()), "p90": q(g, 90),
                              "share_under_1us": float((g < 1).mean()), "share_over_5us": float((g > 5).mean()),
                              "share_negative": float((g < 0).mean())}}
    if step_us:
        out["step_us"] = step_us
        out["workgroups_x_mean_lifetime_over_256_cus_x_step"] = wgs * float(life.mean()) / (256.0 * step_us)
    return out


def measure(X, y, K, chains_roots, packed, depth=8, steps=2500, ring=192, device=0):
    """A context of its own with the stamp ring on (the caller's timed context stays clean), the given current trees, the
    given prepared batches pipelined `depth` deep for `steps` steps -> analyze().  chains_roots: per chain its K tapes."""
    from bsr.device import DeviceContext
    old = os.environ.get("BSR_TILE_STAMPS")
    os.environ["BSR_TILE_STAMPS"] = str(ring)
    try:
        P = len(packed[0][2])
        ctx = DeviceContext(X, y, K=K, n_chains=len(chains_roots), max_batch=P, device=device)
    finally:
        if old is None:
            del os.environ["BSR_TILE_STAMPS"]
        else:
            os.environ["BSR_TILE_STAMPS"] = old
    try:
        for c, tapes in enumerate(chains_roots):
            for k in range(K):
                ctx.set_current(c, k, tapes[k])
            ctx.refresh(c)
        preps = [(ctx.prepare(*r[:5]), np.zeros_like(r[5])) for r in packed]
        tickets = []
        for i in range(200):       # warm-up
            pr, o = preps[i % len(preps)]
            ctx.score_wait_ptr(ctx.score_submit_prepared(pr), o.ctypes.data)
        t0 = time.perf_counter()
        for i in range(steps):
            pr, o = preps[i % len(preps)]
            tickets.append((ctx.score_submit_prepared(pr), o))
            if len(tickets) >= depth:
                t, oo = tickets.pop(0)
                ctx.score_wait_ptr(t, oo.ctypes.data)
        while tickets:
            t, oo = tickets.pop(0)
            ctx.score_wait_ptr(t, oo.ctypes.data)
        el = time.perf_counter() - t0
        st, launches = read_ring(ctx)
        if st is None:
            return {"error": "no stamps (the assembly tile pass k_tile1a writes them; other row passes do not)"}
        res = analyze(st, 1e6 * el / steps)
        res["batches_in_flight"] = depth
        res["tile_launches"] = launches
        return res
    finally:
        ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--chains", type=int, default=0)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4000)
    a = ap.parse_args()
    import json
    from bsr.tape import flatten
    args = argparse.Namespace(batch=a.batch, chains=a.chains, dtype="f64", burnin=300)
    ranks = bench.Ranks()
    wl = bench.build_workload(a.workload, args, ranks)
    bench.generate_batches(wl, 8)
    roots = [[flatten(ch.roots[k]) for k in range(wl["K"])] for ch in wl["chains"]]
    wl["scorer"].close()
    res = measure(wl["X"], wl["y"], wl["K"], roots, wl["packed"], depth=a.depth, steps=a.steps,
                  ring=int(os.environ.get("BSR_TILE_STAMPS", "256")))
    print(json.dumps(res, indent=1))
    if "error" not in res:
        print("CU occupancy by tile-workgroup lifetime, %d batches in flight, %.2f us per step: %.3f of 256 CUs; "
              "gap on a CU median %.2f us (%.0f %% under 1 us, %.0f %% over 5 us)" %
              (a.depth, res["step_us"], res["occupancy_of_256_cus"], res["gap_on_a_cu_us"]["median"],
               100 * res["gap_on_a_cu_us"]["share_under_1us"], 100 * res["gap_on_a_cu_us"]["share_over_5us"]))


if __name__ == "__main__":
    main()
