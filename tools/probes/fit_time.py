import sys, time
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, ".")
import numpy as np
from bsr import BSR
rs = np.random.RandomState(0)
for (N, d, K, MM) in ((1000, 5, 3, 50), (100000, 10, 3, 50)):
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    import pandas as pd
    m = BSR(K, MM)
    t0 = time.perf_counter()
    m.fit(pd.DataFrame(X), pd.Series(y))
    t1 = time.perf_counter()
    p = m.predict(pd.DataFrame(X[:100]))
    print("N=%d d=%d K=%d MM=%d: fit %.2f s, model %s" % (N, d, K, MM, t1 - t0, str(m.model())[:80]))
