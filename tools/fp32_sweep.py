#!/usr/bin/env python3
"""Config 5's tolerance sweep: the same proposals scored by an f64 and an f32 context (f32 storage and tree arithmetic,
f64 accumulation).  Reports |dloglik|/|loglik|, rank-gate flips and accept-decision flips on the real move mix.
Run on the GPU box:  python tools/fp32_sweep.py [--N 1000000 --d 50 --batches 40]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd")); sys.path.insert(0, ROOT)
import numpy as np
from bench import synth
from bsr import proposal as P
from bsr.chain import Chain, DeviceScorer, run_chains

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=1000000); ap.add_argument("--d", type=int, default=50)
ap.add_argument("--K", type=int, default=3); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--batches", type=int, default=40); ap.add_argument("--burnin", type=int, default=400)
a = ap.parse_args()
X, y = synth(a.N, a.d, seed=0)
sc = {dt: DeviceScorer(X, y, a.K, n_chains=1, max_batch=a.batch, dtype=dt) for dt in ("f64", "f32")}
np.random.seed(1000)
ch = Chain(0, sc["f64"], a.N, a.d, a.K, val=10 ** 9)
run_chains([ch], sc["f64"], batch_per_chain=a.batch, max_props=a.burnin)      # a state the chain really visits
for k in range(a.K):                                                          # same current trees in the f32 context
    sc["f32"].ctx.set_current(0, k, ch.tapes[k])
info32 = sc["f32"].ctx.refresh(0)
info64 = sc["f64"].ctx.refresh(0)
N = a.N
def yll(sse, sigma):
    return -sse / (2 * sigma * sigma) - 0.5 * N * np.log(2 * np.pi * sigma * sigma)
rel, n = [], 0
rank_flip = acc_flip = n_acc = n_full = 0
for _ in range(a.batches):
    cands = ch.generate(a.batch)
    ch.rng_state = ch._end_state
    tapes = [c.tape for c in cands]
    ks = np.array([c.k for c in cands], np.int32); sig = np.array([c.new_sigma for c in cands])
    r = {dt: sc[dt].ctx.score_batch(tapes, np.zeros(len(cands), np.int32), ks, sig).copy() for dt in sc}
    for i, c in enumerate(cands):
        n += 1
        full = [int(r[dt]["rank"][i]) == a.K for dt in ("f64", "f32")]
        if full[0] != full[1]:
            rank_flip += 1
            continue
        if not full[0]:
            continue
        n_full += 1
        l64, l32 = float(r["f64"]["loglik"][i]), float(r["f32"]["loglik"][i])
        rel.append(abs(l32 - l64) / abs(l64))
        s_new = P.fstruc_t(c.root, ch.n_feature, ch.T, ch.beta, c.new_sa2, c.new_sb2)
        dec = []
        for ll, info in ((l64, info64), (l32, info32)):
            logR = P.log_ratio(c.change, c.Q, c.Qinv, c.hratio, c.detjacob, ll, yll(info["sse_old"], ch.sigma), s_new,
                               ch._fs_old(c.k), c.new_sigma, ch.sigma)
            dec.append(bool(P.accept_test(logR, c.u)))
        n_acc += dec[0]
        acc_flip += dec[0] != dec[1]
rel = np.array(rel)
print("N=%d d=%d K=%d: %d proposals, %d full rank in both" % (a.N, a.d, a.K, n, n_full))
print("old-state SSE: f64 %.10g  f32 %.10g  (rel %.2e)" % (info64["sse_old"], info32["sse_old"],
      abs(info32["sse_old"] - info64["sse_old"]) / info64["sse_old"]))
print("|dloglik|/|loglik|: median %.2e  p90 %.2e  p99 %.2e  max %.2e" % tuple(np.percentile(rel, [50, 90, 99, 100])))
print("share within 1e-6: %.1f %%, within 1e-5: %.1f %%, within 1e-4: %.1f %%" % tuple(100 * np.mean(rel <= t) for t in (1e-6, 1e-5, 1e-4)))
print("rank-gate flips: %d / %d   accept-decision flips: %d / %d (f64 accepts: %d)" % (rank_flip, n, acc_flip, n_full, n_acc))
