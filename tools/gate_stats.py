import os, sys
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, ".")
import numpy as np
from bench import synth
from bsr.chain import DeviceScorer
from bsr.native import NativeEngine
os.environ["BSR_ENGINE_PREDICT"] = "0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
X, y = synth(100000, 10)
tot = np.zeros((2, 2)); 
for seed in range(1000, 1008):
    sc = DeviceScorer(X, y, K, n_chains=1, max_batch=32)
    eng = NativeEngine(sc.ctx, 1, 10, val=10**9); eng.set_nan_policy(True)
    eng.seed(0, seed); eng.init_chain(0)
    tr = eng.run(batch_per_chain=32, max_props=6000, trace_cap=6100)
    d = (tr["rank"] < K).astype(int); k = tr["count"]
    line = []
    for kk in range(K):
        s = d[k == kk]
        n11 = int(np.sum((s[1:] == 1) & (s[:-1] == 1))); n1 = int(np.sum(s[:-1] == 1))
        n01 = int(np.sum((s[1:] == 1) & (s[:-1] == 0))); n0 = int(np.sum(s[:-1] == 0))
        tot[1, 1] += n11; tot[1, 0] += n1 - n11; tot[0, 1] += n01; tot[0, 0] += n0 - n01
        line.append("k%d %.3f (P(D|D)=%.2f P(D|F)=%.3f)" % (kk, s.mean(), n11 / max(1, n1), n01 / max(1, n0)))
    print(seed, "acc", int(tr["accepted"].sum()), " | ".join(line))
    eng.close(); sc.close()
print("overall P(D|D)=%.3f P(D|F)=%.4f  share D %.4f" % (tot[1,1]/tot[1].sum(), tot[0,1]/tot[0].sum(), tot[:,1].sum()/tot.sum()))
