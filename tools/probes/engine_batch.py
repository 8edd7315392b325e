"""Consumed proposals per second of the native sampler against the speculation length (8 chains, N=100k, d=10)."""
import os, sys, time
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, ".")
import numpy as np
from bench import synth
from bsr.chain import DeviceScorer
from bsr.native import NativeEngine
X, y = synth(100000, 10)
for K in (3, 8):
    for batch in (8, 12, 16, 24, 32, 48):
        chains = 8
        sc = DeviceScorer(X, y, K, n_chains=chains, max_batch=chains * batch)
        eng = NativeEngine(sc.ctx, chains, 10, val=10**9); eng.set_nan_policy(True)
        for c in range(chains):
            eng.seed(c, 1000 + c); eng.init_chain(c)
        eng.run(batch_per_chain=batch, max_props=300)
        n0 = sum(eng.result(c, current=True)["n_props"] for c in range(chains))
        t0 = time.perf_counter()
        eng.run(batch_per_chain=batch, max_props=300 + 6000)
        dt = time.perf_counter() - t0
        r = [eng.result(c, current=True) for c in range(chains)]
        n = sum(x["n_props"] for x in r) - n0; dsc = sum(x["n_discarded"] for x in r)
        print("K=%d batch %2d: %.2f M consumed/s, discarded %.3f" % (K, batch, n / dt / 1e6, dsc / (n + n0 + dsc)), flush=True)
        eng.close(); sc.close()
