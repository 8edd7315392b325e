"""MCMC chain engine: the body of BSR.fit's `while len(trainERRS) < MM` loop (codes/bsr_class.py:99-273) driving
the GPU scorer with speculative batches.

Exactness of speculation: a rejected proposal leaves (sigma, Root, sigma_a, sigma_b) untouched
(codes/funcs.py:1300-1303, codes/bsr_class.py:195-198), so the proposals that follow a rejection are drawn from the
same chain state.  A batch therefore holds the next B proposals of the chain "assuming every one is rejected after
drawing its accept-uniform"; results are then consumed in order and the first event that breaks the assumption ends
the batch:
  * accept              -> adopt the candidate, rewind the RNG to just after that proposal's uniform;
  * rank-gate rejection -> the reference returns BEFORE drawing the uniform (codes/funcs.py:1226-1228), so the RNG
                           is rewound to just before it and the tail of the batch is regenerated -- unless the
                           rejection was speculated: the gate's verdict is mostly a property of the chain state
                           (dependent siblings, siblings 1e14 apart in scale), so a tree whose recent proposals were
                           gate-rejected is speculated as rejected again, no uniform is drawn behind it, and the batch
                           stays on the right stream.  A wrong guess costs the tail, never exactness.
Every chain owns its RNG stream (numpy legacy MT19937 state) so that several chains can share one launch.
"""
import math

import numpy as np

from . import proposal as P
from . import rng
from .node import clone, getNum
from .tape import flatten


class Scorer:
    """What the engine needs from the data side.  The product implementation is DeviceScorer (HIP kernels through the
    C ABI); tests may plug a CPU stand-in to exercise the host logic without a GPU."""

    def set_tree(self, chain, k, tape):
        raise NotImplementedError

    def refresh(self, chain):
        """-> dict(sse_old, colflags)"""
        raise NotImplementedError

    def score(self, tapes, chains, ks, sigmas):
        """-> list/array of records with fields rank, loglik, flags"""
        raise NotImplementedError

    def commit(self, chain, k, slot):
        raise NotImplementedError

    def fit_beta(self, chain):
        """-> (Beta (K+1,1), rmse)"""
        raise NotImplementedError


class DeviceScorer(Scorer):
    """HIP path.  Raises at construction when libbsr_hip.so or a GPU is missing: there is no CPU fallback."""

    def __init__(self, X, y, K, n_chains=1, max_batch=64, device=0, dtype="f64", typical_chains=0, typical_batch=0):
        from .device import DeviceContext
        self.ctx = DeviceContext(X, y, K=K, n_chains=n_chains, max_batch=max_batch, device=device, dtype=dtype,
                                 typical_chains=typical_chains, typical_batch=typical_batch)
        self.K = K
        self.max_batch = max_batch

    def set_tree(self, chain, k, tape):
        self.ctx.set_current(chain, k, tape)

    def refresh(self, chain):
        return self.ctx.refresh(chain)

    def score(self, tapes, chains, ks, sigmas):
        return self.ctx.score_batch(tapes, chains, ks, sigmas)

    def commit(self, chain, k, slot):
        self.ctx.commit(chain, k, slot)

    def fit_beta(self, chain):
        return self.ctx.fit_beta(chain)

    def close(self):
        self.ctx.close()


class _Cand:
    __slots__ = ("k", "root", "tape", "change", "Q", "Qinv", "hratio", "detjacob", "new_sigma", "new_sa2", "new_sb2",
                 "state_before_u", "u", "action", "s_new", "pred_def")


class Chain:
    """One chain: K trees, sigma, per-tree (sigma_a, sigma_b), the sweep position and its private RNG stream."""

    def __init__(self, index, scorer, N, n_feature, K, beta=-1, val=100, table=None, y_is_series=True,
                 rng_state=None, trace=None, feature_range=None):
        self.index = index
        self.scorer = scorer
        self.N = N
        self.n_feature = n_feature
        self.K = K
        self.beta = beta
        self.val = val
        self.T = table or P.default_table()
        self.y_is_series = y_is_series
        self.trace = trace
        self.done = False
        self.n_props = 0
        self.n_accept = 0
        self.n_rank_rejects = 0
        self.n_discarded = 0
        self.def_ema = [0.0] * K           # share of tree k's recent proposals that the rank gate rejected
        self.feature_range = feature_range  # (lo[d], hi[d]) of X, or None: only the history guess is used
        self.colmax = [0.0] * K
        if rng_state is not None:
            rng.set_state(rng_state)
        self._init_state()
        self.rng_state = rng.get_state()

    # -- codes/bsr_class.py:116-163
    def _init_state(self):
        K = self.K
        self.sigma = rng.invgamma_rvs(1)
        self.roots, self.siga, self.sigb = [], [], []
        for k in range(K):
            root = P.Node(0)
            sa = rng.invgamma_rvs(1)
            sb = rng.invgamma_rvs(1)
            P.grow_t(root, self.n_feature, self.T, self.beta, sa, sb)
            self.roots.append(root)
            self.siga.append(sa)
            self.sigb.append(sb)
        self.tapes = [flatten(r) for r in self.roots]
        for k in range(K):
            self.scorer.set_tree(self.index, k, self.tapes[k])
        self._refresh()
        self.Beta, _ = self.scorer.fit_beta(self.index)
        self.total = 0
        self.count = 0            # next tree of the sweep
        self.errs = []
        self.last_roots = list(self.roots)   # `Roots` as built before the latest newProp (codes/bsr_class.py:180-182)
        self.init_roots = list(self.roots)

    def _refresh(self):
        info = self.scorer.refresh(self.index)
        self.colflags = [int(f) for f in info["colflags"]]
        self.colmax = [float(v) for v in info.get("maxabs", [0.0] * self.K)]
        if any(self.colflags):
            # every fitted value of the old state is NaN: Series.sum(skipna=True) gives 0.0, ndarray sum NaN
            self.sse_old = 0.0 if self.y_is_series else float("nan")
        else:
            self.sse_old = float(info["sse_old"])
        self.fs_old = [None] * self.K
        self._ckeys = [None] * self.K

    def _yll(self, sigma):
        # codes/funcs.py:1172-1173 on the cached old-state SSE
        return -self.sse_old / (2 * sigma * sigma) - 0.5 * self.N * math.log(2 * math.pi * sigma * sigma)

    def _fs_old(self, k):
        if self.fs_old[k] is None:
            self.fs_old[k] = P.fstruc_t(self.roots[k], self.n_feature, self.T, self.beta, self.siga[k], self.sigb[k])
        return self.fs_old[k]

    def _ckey(self, j):
        if self._ckeys[j] is None:
            self._ckeys[j] = P.canon_key(self.roots[j])
        return self._ckeys[j]

    # -- rank-gate guess (csrc/bsr_engine.hip: predict_gate_reject has the same rules) ---------------------------
    def _predict_reject(self, root, k):
        if self.def_ema[k] > 0.9:
            return True
        if self.K < 2:
            return False
        key = P.canon_key(root)
        sibs = [self._ckey(j) for j in range(self.K) if j != k]
        if key in sibs or len(set(sibs)) < len(sibs):                # repeats a sibling / siblings repeat each other
            return True
        if self.feature_range is None:
            return False
        if any(self.colflags[j] for j in range(self.K) if j != k):
            return True
        sib = max(self.colmax[j] for j in range(self.K) if j != k)
        lo, hi = P.tree_range(root, self.feature_range[0], self.feature_range[1], float(self.N))
        est = max(abs(lo), abs(hi))
        if est != est:
            return False
        if math.isinf(est):
            return True
        tol = max(self.N, self.K) * 2.220446049250313e-16
        return est > sib * (10.0 / tol) or est < sib * (tol / 10.0)

    # -- speculative generation ---------------------------------------------------------------------------------
    def generate(self, max_n):
        """Draws up to max_n proposals from the current state under the all-rejected assumption."""
        rng.set_state(self.rng_state)
        cands = []
        total, count = self.total, self.count
        while len(cands) < max_n:
            if count == 0 and total >= self.val:     # `while total < val` is only tested between sweeps
                break
            k = count
            c = _Cand()
            c.k = k
            mv = P.prop_inplace(clone(self.roots[k]), self.n_feature, self.T, self.beta, self.siga[k], self.sigb[k])
            c.new_sigma = rng.invgamma_rvs(P.SIG_SHAPE)
            c.new_sa2, c.new_sb2, c.hratio, c.detjacob = P.aux_inplace(
                mv.change, mv.root, mv.ln_nodes, self.siga[k], self.sigb[k], mv.last_a, mv.last_b)
            c.root, c.change, c.Q, c.Qinv, c.action = mv.root, mv.change, mv.Q, mv.Qinv, mv.action
            c.tape = flatten(c.root)
            c.state_before_u = rng.get_state()
            c.pred_def = self._predict_reject(mv.root, k)
            c.u = None if c.pred_def else rng.uniform()
            cands.append(c)
            total += 1
            count = (count + 1) % self.K
        self._cands = cands
        self._end_state = rng.get_state()
        return cands

    # -- consumption ----------------------------------------------------------------------------------------------
    def consume(self, results, slots):
        """Applies the scored candidates in order; returns the number of proposals that were really consumed."""
        used = 0
        broke = False
        for c, res, slot in zip(self._cands, results, slots):
            used += 1
            self.n_props += 1
            k = c.k
            self.last_roots = list(self.roots)
            rank = int(res["rank"])
            rec = None
            if self.trace is not None:
                rec = {"count": k, "action": c.action, "change": c.change, "Q": c.Q, "Qinv": c.Qinv,
                       "new_sigma": c.new_sigma, "new_sa2": c.new_sa2, "new_sb2": c.new_sb2, "rank": rank,
                       "proposed": c.root, "accepted": False, "cur_roots": self.last_roots}
            if rank < 0:
                self.rng_state = c.state_before_u
                raise np.linalg.LinAlgError("SVD did not converge")   # NaN in new_outputs, codes/funcs.py:1226
            self.total += 1
            self.count = (k + 1) % self.K
            if rank < self.K:                                          # codes/funcs.py:1226-1228: no uniform drawn
                self.n_rank_rejects += 1
                self.def_ema[k] = 0.75 * self.def_ema[k] + 0.25
                if rec is not None:
                    self.trace(rec)
                if c.pred_def:                                         # speculated exactly that: the batch goes on
                    continue
                self.rng_state = c.state_before_u
                broke = True
                break
            self.def_ema[k] *= 0.75
            tail_invalid = False
            if c.pred_def:                                             # passed the gate after all: draw its uniform now
                rng.set_state(c.state_before_u)
                c.u = rng.uniform()
                self.rng_state = rng.get_state()
                tail_invalid = True
            yllstar = float(res["loglik"])
            yll = self._yll(self.sigma)
            s_new = P.fstruc_t(c.root, self.n_feature, self.T, self.beta, c.new_sa2, c.new_sb2)
            logR = P.log_ratio(c.change, c.Q, c.Qinv, c.hratio, c.detjacob, yllstar, yll, s_new, self._fs_old(k),
                               c.new_sigma, self.sigma)
            accepted = P.accept_test(logR, c.u)
            if rec is not None:
                rec.update(yllstar=yllstar, yll=yll, logR=logR, accepted=accepted, u=c.u)
            if not accepted:
                if rec is not None:
                    self.trace(rec)
                if tail_invalid:                                       # what follows was drawn on a shifted stream
                    broke = True
                    break
                continue
            # ---- accepted: codes/bsr_class.py:200-243
            self.n_accept += 1
            self.roots[k] = c.root
            self.tapes[k] = c.tape
            self.sigma = c.new_sigma
            self.siga[k] = c.new_sa2
            self.sigb[k] = c.new_sb2
            self.scorer.commit(self.index, k, slot)
            self._refresh()
            self.Beta, rmse = self.scorer.fit_beta(self.index)
            self.errs.append(rmse)
            self.total = 0
            for j in range(self.K):
                if j != k:
                    self.def_ema[j] = 0.0                              # their sibling set has changed
            rng.set_state(c.state_before_u)
            rng.uniform()
            self.rng_state = rng.get_state()
            if rec is not None:
                rec["rmse"] = rmse
                self.trace(rec)
            m = min(10, len(self.errs))                                # codes/bsr_class.py:248-252
            if len(self.errs) > 100 and 1 - min(self.errs[-m:]) / (sum(self.errs[-m:]) / m) < 0.05:
                self.done = True
            broke = True
            break
        if not broke:
            self.rng_state = self._end_state       # every candidate was consumed as a plain rejection
        self.n_discarded += len(self._cands) - used
        if not self.done and self.count == 0 and self.total >= self.val:
            self.done = True
        self._cands = []
        return used

    def result(self):
        """What BSR.fit appends per chain (codes/bsr_class.py:270-273): the `Roots` list built before the last
        newProp (stale by one accept when the plateau rule fired), the last Beta and the per-accept RMSE list."""
        return {"roots": self.last_roots, "beta": self.Beta, "errs": self.errs, "n_props": self.n_props,
                "n_accept": self.n_accept, "complexity": sum(getNum(r) for r in self.last_roots)}


def run_chains(chains, scorer, batch_per_chain=32, max_props=None):
    """Advances the chains in lock-step launches until all are done.  Each launch carries up to batch_per_chain
    speculative proposals of every live chain (sum bounded by the scorer's max_batch)."""
    while True:
        live = [c for c in chains if not c.done]
        if max_props is not None:
            live = [c for c in live if c.n_props < max_props]
        if not live:
            break
        cap = max(1, scorer.max_batch // len(live)) if hasattr(scorer, "max_batch") else batch_per_chain
        per = min(batch_per_chain, cap)
        tapes, chs, ks, sig, owner = [], [], [], [], []
        for c in live:
            room = per if max_props is None else min(per, max_props - c.n_props)
            cands = c.generate(room)
            base = len(tapes)
            for cd in cands:
                tapes.append(cd.tape)
                chs.append(c.index)
                ks.append(cd.k)
                sig.append(cd.new_sigma)
            owner.append((c, base, len(cands)))
            if not cands:
                c.done = True
        if not tapes:
            continue
        res = scorer.score(tapes, chs, ks, sig)
        for c, base, n in owner:
            if n:
                c.consume(res[base:base + n], range(base, base + n))
    return chains
