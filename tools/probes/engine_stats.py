"""Where the native sampler's speculation is lost: accepted moves and gate mis-predictions per proposal, K = 3 and 8."""
import os, sys
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, ".")
import numpy as np
from bench import synth
from bsr.chain import DeviceScorer
from bsr.native import NativeEngine
X, y = synth(100000, 10)
for K in (3, 8):
    for batch in (16, 32):
        chains = 8
        sc = DeviceScorer(X, y, K, n_chains=chains, max_batch=chains * batch)
        eng = NativeEngine(sc.ctx, chains, 10, val=10**9); eng.set_nan_policy(True)
        for c in range(chains):
            eng.seed(c, 1000 + c); eng.init_chain(c)
        eng.run(batch_per_chain=batch, max_props=4000)
        r = [eng.result(c, current=True) for c in range(chains)]
        n = sum(x["n_props"] for x in r); a = sum(x["n_accept"] for x in r); rr = sum(x["n_rank_rejects"] for x in r)
        dsc = sum(x["n_discarded"] for x in r)
        print("K=%d batch %d: consumed %d, accepted %.3f, gate-rejected %.3f, discarded/(consumed+discarded) %.3f, discarded per accept %.1f"
              % (K, batch, n, a / n, rr / n, dsc / (n + dsc), dsc / max(1, a)))
        eng.close(); sc.close()
