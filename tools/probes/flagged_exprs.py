"""What the candidates of the real move mix that take the residual step look like: columns (nearly) inside the span of the
chain's K current columns (rho^2 < 1e-6 |z|^2), self-repeats aside.  Host arithmetic on evaluated columns."""
import os
import sys
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, ".")
import numpy as np
from bench import synth
from bsr.chain import Chain, DeviceScorer, run_chains
from bsr import proposal as P
from bsr.node import Express
from bsr.tape import flatten
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
# the library's own structure analysis (csrc/bsr_span.h) through the test shim: what it does NOT recognise is what
# still goes through the residual step
import ctypes as C, subprocess, tempfile
_so = os.path.join(tempfile.mkdtemp(), "libspan.so")
subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", _so, "tests/native/span_shim.cpp"], check=True)
_L = C.CDLL(_so)
_L.span_check.restype = C.c_int
_L.span_check.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
from bsr.tape import pack
def recognised(cur_tapes, cand_tapes, ks):
    rows, off = pack(list(cur_tapes) + list(cand_tapes))
    off = np.ascontiguousarray(off, dtype=np.int32)
    ks = np.ascontiguousarray(ks, dtype=np.int32)
    out = np.zeros(len(cand_tapes), dtype=np.int32)
    _L.span_check(rows.ctypes.data, off.ctypes.data, len(cur_tapes), len(cand_tapes), ks.ctypes.data, out.ctypes.data)
    return out
import os
NN, DD = int(os.environ.get("FE_N", 100000)), int(os.environ.get("FE_D", 10))
X, y = synth(NN, DD)
sc = DeviceScorer(X, y, K, n_chains=1, max_batch=72)
np.random.seed(1000)
ch = Chain(0, sc, NN, DD, K, val=10 ** 9)
run_chains([ch], sc, batch_per_chain=32, max_props=300)
shown = n = n_amb = n_self = nb_amb = 0
for b in range(40):
    cands = ch.generate(64)
    cur = sc.ctx.eval_tapes([flatten(r) for r in ch.roots])[0]       # [K][N]
    Q, _ = np.linalg.qr(np.asarray(cur).T)
    Z = np.asarray(sc.ctx.eval_tapes([c.tape for c in cands])[0])
    keys = [ch._ckey(j) for j in range(K)]
    rec = recognised([flatten(r) for r in ch.roots], [c.tape for c in cands], [c.k for c in cands])
    any_amb = False
    for i, c in enumerate(cands):
        z = Z[i]
        n += 1
        if not np.isfinite(z).all():
            continue
        zz = float(z @ z)
        cc = Q.T @ z
        rho2 = zz - float(cc @ cc)
        slf = rec[i] >= 1
        if zz == 0 or rho2 <= 1e-6 * zz:
            n_amb += 1
            n_self += slf
            if not slf:
                any_amb = True
                if shown < 40:
                    shown += 1
                    print("k=%d cand: %s | zz %.3g rho2/zz %.3g" % (c.k, Express(c.root), zz, rho2 / zz if zz else 0))
                    print("      current:", " ;; ".join(Express(ch.roots[j]) for j in range(K)))
    nb_amb += any_amb
    ch.rng_state = ch._end_state
print("proposals %d, in span %d (%.3f), of them self-repeats %d; batches with one that is not: %d of 40" % (n, n_amb, n_amb / n, n_self, nb_amb))
sc.close()
