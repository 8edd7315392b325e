"""Module-level functions of the reference's `bsr.funcs`, same names and argument meaning
(codes/funcs.py; re-exported by codes/__init__.py:9-11).  The data-parallel ones run on the GPU:

  allcal(node, indata)            codes/funcs.py:175-220   -> HIP stack-machine kernel
  ylogLike(y, outputs, sigma)     codes/funcs.py:1147-1174 -> HIP Gram/solve/residual kernels
  newProp(Roots, count, ...)      codes/funcs.py:1184-1306 -> host proposal + one scored candidate on the GPU

There is no CPU path: without libbsr_hip.so and an MI355X these raise.
"""
import weakref

import numpy as np

from . import proposal as P
from . import rng
from .node import (Node, Operator, genList, shrink, upgOd, display, getHeight, getNum, numLT, upDepth, Express,
                   clone)
from .tape import flatten


def _table(Ops, Op_weights, Op_type):
    return P.OpTable.get(Ops, Op_weights, Op_type)


def grow(node, nfeature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b):
    """codes/funcs.py:74-119"""
    P.grow_t(node, nfeature, _table(Ops, Op_weights, Op_type), beta, sigma_a, sigma_b)


def fStruc(node, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b):
    """codes/funcs.py:349-398 -> [loglike, loglike_para]"""
    return list(P.fstruc_t(node, n_feature, _table(Ops, Op_weights, Op_type), beta, sigma_a, sigma_b))


def Prop(Root, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b):
    """codes/funcs.py:406-923 -> [oldRoot, Root, lnPointers, change, Q, Qinv, last_a, last_b, cnode]"""
    oldRoot = clone(Root)
    mv = P.prop_inplace(Root, n_feature, _table(Ops, Op_weights, Op_type), beta, sigma_a, sigma_b)
    return [oldRoot, mv.root, mv.ln_nodes, mv.change, mv.Q, mv.Qinv, mv.last_a, mv.last_b, mv.cnode]


def auxProp(change, oldRoot, Root, lnPointers, sigma_a, sigma_b, last_a, last_b, cnode=None):
    """codes/funcs.py:935-1138 -> [hratio, detjacob, new_sa2, new_sb2] or [new_sa2, new_sb2]"""
    sa2, sb2, h, dj = P.aux_inplace(change, Root, lnPointers, sigma_a, sigma_b, last_a, last_b)
    if change in ('shrinkage', 'expansion'):
        return [h, dj, sa2, sb2]
    return [sa2, sb2]


# ----------------------------------------------------------------------------------------------------------------
class _Contexts:
    """Small cache of device contexts keyed by the identity of the caller's data objects."""

    def __init__(self, limit=4):
        self.limit = limit
        self.items = []

    @staticmethod
    def _alive(ref, obj):
        return ref is not None and ref() is obj

    @staticmethod
    def _ref(obj):
        try:
            return weakref.ref(obj)
        except TypeError:
            return None

    def get(self, indata, y, K, build):
        for it in self.items:
            if it["K"] == K and self._alive(it["x"], indata) and (y is None and it["y"] is None or
                                                                   self._alive(it["y"], y)):
                return it["obj"]
        obj = build()
        self.items.append({"K": K, "x": self._ref(indata), "y": None if y is None else self._ref(y), "obj": obj})
        if len(self.items) > self.limit:
            old = self.items.pop(0)
            try:
                old["obj"].close()
            except Exception:
                pass
        return obj


_eval_ctx = _Contexts()
_score_ctx = _Contexts()


def _as_matrix(indata):
    return np.ascontiguousarray(np.asarray(indata, dtype=np.float64))


def allcal(node, indata):
    """Evaluates the tree on every row of `indata` -> (N,1) float64 array (codes/funcs.py:175-220).
    Integer inputs are cast to float64 (the reference would truncate exp/inv in place on integer frames)."""
    from .device import DeviceContext
    if node.type == -1:
        print("Not a grown tree!")
        return node.data
    ctx = _eval_ctx.get(indata, None, 0, lambda: DeviceContext(_as_matrix(indata), None, max_batch=16))
    cols, _, _ = ctx.eval_tapes([flatten(node)])
    node.data = cols[0].reshape(-1, 1)
    return node.data


def _is_series(y):
    return hasattr(y, "iloc")


def ylogLike(y, outputs, sigma):
    """codes/funcs.py:1147-1174"""
    from .device import yloglike_device
    return yloglike_device(y, outputs, sigma, skipna=_is_series(y))["loglik"]


class _PropState:
    """Scorer + the tapes currently loaded for the K trees of the caller's `Roots`."""

    def __init__(self, indata, y, K):
        from .chain import DeviceScorer
        self.scorer = DeviceScorer(_as_matrix(indata), np.asarray(y, dtype=np.float64), K, n_chains=1, max_batch=4)
        self.loaded = [None] * K
        self.info = None

    def close(self):
        self.scorer.close()

    def sync(self, Roots):
        dirty = False
        for k, r in enumerate(Roots):
            t = flatten(r)
            if self.loaded[k] is None or self.loaded[k].tobytes() != t.tobytes():
                self.scorer.set_tree(0, k, t)
                self.loaded[k] = t
                dirty = True
        if dirty or self.info is None:
            self.info = self.scorer.refresh(0)
        return self.info


def newProp(Roots, count, sigma, y, indata, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b):
    """One Metropolis-Hastings proposal on tree `count` (codes/funcs.py:1184-1306).
    Returns [accepted, sigma, Root, sigma_a, sigma_b] like the reference (Root is a fresh copy)."""
    import math
    K = len(Roots)
    T = _table(Ops, Op_weights, Op_type)
    st = _score_ctx.get(indata, y, K, lambda: _PropState(indata, y, K))
    info = st.sync(Roots)
    N = len(y)
    oldRoot = Roots[count]
    mv = P.prop_inplace(clone(oldRoot), n_feature, T, beta, sigma_a, sigma_b)
    new_sigma = rng.invgamma_rvs(P.SIG_SHAPE)
    new_sa2, new_sb2, hratio, detjacob = P.aux_inplace(mv.change, mv.root, mv.ln_nodes, sigma_a, sigma_b,
                                                       mv.last_a, mv.last_b)
    res = st.scorer.score([flatten(mv.root)], [0], [count], [new_sigma])[0]
    rank = int(res["rank"])
    if rank < 0:
        raise np.linalg.LinAlgError("SVD did not converge")
    if rank < K:
        return [False, sigma, clone(oldRoot), sigma_a, sigma_b]
    if any(int(f) for f in info["colflags"]):
        sse_old = 0.0 if _is_series(y) else float("nan")
    else:
        sse_old = float(info["sse_old"])
    yll = -sse_old / (2 * sigma * sigma) - 0.5 * N * math.log(2 * math.pi * sigma * sigma)
    s_new = P.fstruc_t(mv.root, n_feature, T, beta, new_sa2, new_sb2)
    s_old = P.fstruc_t(oldRoot, n_feature, T, beta, sigma_a, sigma_b)
    logR = P.log_ratio(mv.change, mv.Q, mv.Qinv, hratio, detjacob, float(res["loglik"]), yll, s_new, s_old,
                       new_sigma, sigma)
    u = rng.uniform()
    if not P.accept_test(logR, u):
        return [False, sigma, clone(oldRoot), sigma_a, sigma_b]
    return [True, new_sigma, clone(mv.root), new_sa2, new_sb2]
