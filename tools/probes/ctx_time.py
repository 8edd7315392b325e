import sys, time, os
sys.path.insert(0, 'mcmc-symreg_amd')
import numpy as np
from bsr.device import DeviceContext
rs = np.random.RandomState(0)
for N, d in ((500, 4), (5000, 10), (100000, 10)):
    X = rs.uniform(-3, 3, size=(N, d)); y = rs.standard_normal(N)
    t0 = time.perf_counter()
    for i in range(5):
        c = DeviceContext(X, y, K=3, n_chains=1, max_batch=64)
        c.close()
    print(os.environ.get("BSR_DERIVED"), N, d, "ctx create+close %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
