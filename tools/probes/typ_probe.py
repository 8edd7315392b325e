"""Probe: narrow (two chains, 64 proposals) against wide (eight chains, 256) batches on one context, byte for byte.
  python tools/probes/typ_probe.py [tuned]   -- tuned: bsr_ctx_create_tuned(typical_chains=2, typical_batch=64)"""
import os, sys
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, "tests")
import numpy as np
from bsr.device import DeviceContext
from test_gpu_dispatch import _trees
rs = np.random.RandomState(4)
N, d, K, C = 100_000, 10, 3, 8
X = rs.uniform(-3, 3, size=(N, d)); y = X[:, 0] * X[:, 1] + np.sin(X[:, 2]) + 0.1 * rs.standard_normal(N)
tuned = len(sys.argv) > 1 and sys.argv[1] == 'tuned'
ctx = DeviceContext(X, y, K=K, n_chains=C, max_batch=256, typical_chains=2 if tuned else 0, typical_batch=64 if tuned else 0)
print(ctx.info())
for c in range(C):
    cur = _trees(d, rs, 7 * K)[:K]
    for k in range(K):
        ctx.set_current(c, k, cur[k])
    ctx.refresh(c)
tapes = _trees(d, rs, 256)
chains = (np.arange(256) // 32).astype(np.int32)
ks = rs.randint(K, size=256).astype(np.int32)
sig = rs.uniform(0.5, 1.5, size=256)
wide = ctx.score_batch(tapes, chains, ks, sig)
bad = 0
for g in range(4):
    sel = np.arange(g * 64, (g + 1) * 64)
    nar = ctx.score_batch([tapes[i] for i in sel], chains[sel], ks[sel], sig[sel])
    bad += sum(nar[i].tobytes() != wide[sel[i]].tobytes() for i in range(64))
print("narrow vs wide batches differ in", bad, "of 256; ranks", np.bincount(wide["rank"].clip(0)), ctx.dispatch_info())
ctx.close()
