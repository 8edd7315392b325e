#!/usr/bin/env python3
"""End-to-end chain throughput (consumed MH proposals per second, host proposal generation included)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, ROOT)
import numpy as np
from bench import synth
from bsr.chain import Chain, DeviceScorer, run_chains
from bsr.native import NativeEngine

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=100000); ap.add_argument("--d", type=int, default=10)
ap.add_argument("--K", type=int, default=3); ap.add_argument("--chains", type=int, default=8)
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--props", type=int, default=2000)
ap.add_argument("--engine", default="native")
a = ap.parse_args()
X, y = synth(a.N, a.d)
sc = DeviceScorer(X, y, a.K, n_chains=a.chains, max_batch=a.batch * a.chains)
t0 = time.perf_counter()
if a.engine == "native":
    eng = NativeEngine(sc.ctx, a.chains, a.d, val=10 ** 9)
    eng.set_nan_policy(True)
    for c in range(a.chains):
        eng.seed(c, 1000 + c); eng.init_chain(c)
    t1 = time.perf_counter()
    eng.run(batch_per_chain=a.batch, max_props=a.props)
    t2 = time.perf_counter()
    res = [eng.result(c, current=True) for c in range(a.chains)]
    props = sum(r["n_props"] for r in res); acc = sum(r["n_accept"] for r in res)
    rej = sum(r["n_rank_rejects"] for r in res); disc = sum(r["n_discarded"] for r in res)
else:
    chains = []
    for c in range(a.chains):
        np.random.seed(1000 + c); chains.append(Chain(c, sc, a.N, a.d, a.K, val=10 ** 9))
    t1 = time.perf_counter()
    run_chains(chains, sc, batch_per_chain=a.batch, max_props=a.props)
    t2 = time.perf_counter()
    props = sum(c.n_props for c in chains); acc = sum(c.n_accept for c in chains)
    rej = sum(c.n_rank_rejects for c in chains); disc = sum(c.n_discarded for c in chains)
print("engine %s N=%d d=%d K=%d chains=%d batch=%d: %d proposals consumed in %.3f s = %.0f proposals/s  "
      "(accepts %d, rank-gate rejects %d, speculated-and-discarded %d, init %.3f s)"
      % (a.engine, a.N, a.d, a.K, a.chains, a.batch, props, t2 - t1, props / (t2 - t1), acc, rej, disc, t1 - t0))
