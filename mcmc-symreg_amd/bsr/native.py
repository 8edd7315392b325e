"""NativeEngine: the C++ sampler of libbsr_hip.so (csrc/bsr_engine.hip) behind a small Python handle.

Same algorithm and random-draw order as bsr.chain / bsr.proposal (and therefore as the reference), but the whole
chain loop -- propose, score on the GPU, accept/reject, refit -- runs in native code.  RNG streams are numpy legacy
RandomState streams: seed them per chain (`seed(chain, s)` == np.random.seed(s)) or hand over numpy's global state.
"""
import ctypes as C

import numpy as np

from . import _lib
from .tape import NODE_DTYPE, unflatten

MAX_RECORD_NODES = 4096


def default_batch(N, d, K):
    """Speculative proposals per chain and batch where the caller names none (BSR(batch=None), fit_sharded, run_rank).
    A batch is drawn on the assumption that every proposal in it is rejected (codes/funcs.py:1300-1303: what is behind the
    first accept or unforeseen gate verdict is thrown away), so a deeper batch buys fewer, fuller launches with more
    discarded scores.  Where a launch's fixed costs dominate -- data sets whose row slices sit in LDS whole -- 64 pays
    (8 chains, N=100k, d=10: 5.4 M consumed proposals/s against 4.9 M at 32, K=8: +4 %, a lone chain +10-20 %;
    profiles/r06_engine_batch_ab.txt); where the row pass itself is the bound (N=1M, d=50: 0.23 against 0.27 M/s) every
    discarded score costs, and 32 stays."""
    return 64 if int(N) * (int(d) + int(K) + 1) <= 8_000_000 else 32


def batch_shape(n_chains, batch_per_chain, K):
    """(typical_chains, typical_batch) of the batches the native sampler submits for `n_chains` chains: its worker threads
    take the chains in up to four groups (csrc/bsr_engine.hip: bsr_engine_run; BSR_ENGINE_GROUPS) and a batch holds one
    group's proposals.  For DeviceScorer / DeviceContext, so that the row pass's geometry fits THOSE batches.  K == 1 and
    traced runs submit every chain at once: the context's limits then."""
    import os
    if K <= 1 or n_chains <= 1:
        return 0, 0
    groups = max(1, min(8, int(os.environ.get("BSR_ENGINE_GROUPS", "4") or 4), n_chains))
    per = -(-n_chains // groups)
    return per, per * int(batch_per_chain)


class NativeEngine:
    def __init__(self, ctx, n_chains, n_feature, beta=-1, val=100, y_is_series=True):
        self._L = _lib.lib()
        self.ctx = ctx
        self.K = ctx.K
        self.n_chains = n_chains
        self._h = C.c_void_p()
        rc = self._L.bsr_engine_create(C.byref(self._h), ctx._h, n_chains, ctx.K, ctx.N, n_feature, float(beta),
                                       int(val), 1 if y_is_series else 0)
        _lib.check(rc, ctx._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.bsr_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == -8:
            raise np.linalg.LinAlgError("SVD did not converge")   # NaN reached the rank gate, codes/funcs.py:1226
        if rc != 0:
            msg = self._L.bsr_engine_last_error(self._h)
            raise _lib.BsrError(rc, msg.decode() if msg else "")

    def set_nan_policy(self, reject):
        """False (default): NaN candidates raise LinAlgError like the reference; True: they are rejected."""
        self._check(self._L.bsr_engine_set_nan_policy(self._h, 1 if reject else 0))

    def set_ops(self, ops, weights):
        """Operator table and prior weights (names as in bsr.node.OP_CODE); default: the reference's ten, uniform."""
        from .node import OP_CODE
        codes = np.array([OP_CODE[o] for o in ops], dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if len(codes) != len(w):
            raise ValueError("ops and op_weights differ in length")
        self._check(self._L.bsr_engine_set_ops(self._h, len(codes), _lib.ptr(codes), _lib.ptr(w)))

    def seed(self, chain, seed):
        self._check(self._L.bsr_engine_seed(self._h, chain, int(seed) & 0xFFFFFFFF))

    def set_numpy_state(self, chain, state=None):
        st = np.random.get_state() if state is None else state
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        self._check(self._L.bsr_engine_set_rng(self._h, chain, _lib.ptr(key), int(st[2]), int(st[3]), float(st[4])))

    def get_numpy_state(self, chain):
        key = np.zeros(624, dtype=np.uint32)
        pos, hg, g = C.c_int32(), C.c_int32(), C.c_double()
        self._check(self._L.bsr_engine_get_rng(self._h, chain, _lib.ptr(key), C.byref(pos), C.byref(hg), C.byref(g)))
        return ("MT19937", key, pos.value, hg.value, g.value)

    def init_chain(self, chain):
        self._check(self._L.bsr_engine_init_chain(self._h, chain))

    def run(self, batch_per_chain=32, max_props=-1, trace_cap=0):
        """Runs every initialised chain to completion (or max_props).  Returns the trace records if requested."""
        trace = np.zeros(max(trace_cap, 1), dtype=_lib.TRACE_DTYPE)
        n = C.c_int64(0)
        rc = self._L.bsr_engine_run(self._h, int(batch_per_chain), int(max_props),
                                    _lib.ptr(trace) if trace_cap else None, int(trace_cap), C.byref(n),
                                    int(self.ctx.max_batch))
        self._check(rc)
        return trace[:n.value]

    def memo_stats(self, chain):
        """(proposals answered from the chain's score memo, lookups) over the chain's life (bsr_engine_memo_stats)."""
        v = np.zeros(2, dtype=np.int64)
        self._check(self._L.bsr_engine_memo_stats(self._h, chain, _lib.ptr(v)))
        return int(v[0]), int(v[1])

    def result(self, chain, current=False):
        K = self.K
        cap = 256
        while True:
            tapes = np.zeros((K, cap), dtype=NODE_DTYPE)
            lens = np.zeros(K, dtype=np.int32)
            beta = np.zeros(K + 1)
            n_errs = C.c_int32(0)
            errs = np.zeros(1 << 16)
            counters = np.zeros(5, dtype=np.int64)
            sigma = C.c_double(0)
            rc = self._L.bsr_engine_chain_result(self._h, chain, _lib.ptr(tapes), cap, _lib.ptr(lens), _lib.ptr(beta),
                                                 _lib.ptr(errs), errs.size, C.byref(n_errs), _lib.ptr(counters),
                                                 C.byref(sigma), 1 if current else 0)
            self._check(rc)
            if (lens >= 0).all() or cap >= 65536:
                break
            cap *= 4
        roots = [unflatten(tapes[k, :lens[k]]) for k in range(K)]
        return {"roots": roots, "beta": beta.reshape(-1, 1), "errs": [float(v) for v in errs[:n_errs.value]],
                "n_props": int(counters[0]), "n_accept": int(counters[1]), "n_rank_rejects": int(counters[2]),
                "n_discarded": int(counters[3]), "done": bool(counters[4]), "sigma": sigma.value,
                "tapes": [tapes[k, :lens[k]].copy() for k in range(K)]}
