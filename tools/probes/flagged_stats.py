"""Which proposals of the real move mix are rank-deficient, and how many of those repeat a current tree structurally
(same canonical key as a sibling, or as the tree they replace)."""
import sys
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, ".")
import numpy as np
from bench import synth
from bsr.chain import Chain, DeviceScorer, run_chains
from bsr import proposal as P
for K in (3, 8):
    X, y = synth(100000, 10)
    sc = DeviceScorer(X, y, K, n_chains=1, max_batch=72)
    np.random.seed(1000)
    ch = Chain(0, sc, 100000, 10, K, val=10 ** 9)
    run_chains([ch], sc, batch_per_chain=32, max_props=300)
    n = n_def = n_sib = n_self = n_def_sib = 0
    for b in range(30):
        cands = ch.generate(64)
        res = sc.ctx.score_batch([c.tape for c in cands], [0] * 64, [c.k for c in cands], [c.new_sigma for c in cands])
        keys = [ch._ckey(j) for j in range(K)]
        for i, c in enumerate(cands):
            key = P.canon_key(c.root)
            sib = any(key == keys[j] for j in range(K) if j != c.k)
            slf = key == keys[c.k]
            dfc = int(res["rank"][i]) < K
            n += 1; n_def += dfc; n_sib += sib; n_self += slf; n_def_sib += (dfc and sib)
        ch.rng_state = ch._end_state
    print("K=%d: %d proposals, rank-deficient %.3f, repeats a sibling %.3f (of the deficient: %.2f), repeats its own tree %.3f"
          % (K, n, n_def / n, n_sib / n, n_def_sib / max(1, n_def), n_self / n))
    sc.close()
