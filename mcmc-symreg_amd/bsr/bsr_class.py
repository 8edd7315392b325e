"""BSR estimator: the reference's public class (codes/bsr_class.py:26-278) on top of the GPU scorer.

Constructor arguments, fit/predict/model/complexity and the fitted attributes roots_/betas_/train_err_ keep the
reference's meaning.  Extra keyword-only options (defaults preserve the reference's behaviour):
  device, dtype          GPU index and compute type ("f64" | "f32")
  batch                  speculative proposals per launch and chain (None: bsr.native.default_batch -- 64 where a row slice
                         sits in LDS whole, 32 for data sets that stream)
  engine                 "native": the C++ sampler of libbsr_hip.so drives the GPU (default); "python": the
                         bsr.chain / bsr.proposal implementation of the same algorithm (same results)
  chain_seeds            None: chains run one after the other on the global numpy RNG stream exactly like the
                         reference; a list of ints: chain c is seeded with chain_seeds[c] and chains advance
                         together, several per launch (independent restarts are the reference's only parallelism,
                         codes/bsr_class.py:99)
  ops, op_weights        operator table and prior weights (the reference hard-codes them, codes/bsr_class.py:110-112).
                         Default: the reference's ten operators with weight 1/10 each.  Names from bsr.node.OP_CODE:
                         the ten plus the extensions 'sub', 'div', 'log'.  Weights need not sum to one.
  devices                None: everything runs in this process on `device`.  A list of GPU indices: the chains are
                         sharded over one fresh process per listed GPU (chain c -> devices[c % len], seeded
                         chain_seeds[c], default 1000 + c), each rank drives its share with the native sampler and
                         the outcomes are gathered over RCCL (bsr.sharded; SURVEY 8e, BASELINE configs[3])
"""
import numpy as np

from . import proposal as P
from . import rng
from .chain import Chain, DeviceScorer, run_chains
from .node import Express, getNum

try:  # sklearn is optional on the GPU box
    from sklearn.base import BaseEstimator, RegressorMixin
except Exception:  # pragma: no cover
    class BaseEstimator(object):
        pass

    class RegressorMixin(object):
        pass


class BSR(BaseEstimator, RegressorMixin):
    def __init__(self, treeNum=3, itrNum=5000, alpha1=0.4, alpha2=0.4, beta=-1, disp=False, val=100,
                 device=0, dtype="f64", batch=None, chain_seeds=None, chains_per_launch=8, engine="native",
                 devices=None, ops=None, op_weights=None):
        self.treeNum = treeNum
        self.itrNum = itrNum
        self.alpha1 = alpha1
        self.alpha2 = alpha2
        self.beta = beta
        self.disp = disp
        self.val = val
        self.device = device
        self.dtype = dtype
        self.batch = batch
        self.chain_seeds = chain_seeds
        self.chains_per_launch = chains_per_launch
        self.engine = engine
        self.devices = devices
        self.ops = ops
        self.op_weights = op_weights

    # ---- codes/bsr_class.py:37-51
    def model(self, last_ind=1):
        return [Express(self.roots_[-last_ind][i]) for i in range(self.treeNum)]

    def complexity(self):
        return sum(getNum(self.roots_[-1][i]) for i in range(self.treeNum))

    # ---- codes/bsr_class.py:53-68
    def predict(self, test_data, method='last', last_ind=1):
        from .device import DeviceContext
        from .tape import flatten
        X = np.ascontiguousarray(np.asarray(test_data, dtype=np.float64))
        K = self.treeNum
        if method != 'last':
            raise UnboundLocalError("local variable 'toutput' referenced before assignment")  # codes/bsr_class.py:59,68
        ctx = DeviceContext(X, None, max_batch=max(K, 1), device=self.device, dtype=self.dtype)
        try:
            cols, _, _ = ctx.eval_tapes([flatten(self.roots_[-last_ind][k]) for k in range(K)])
        finally:
            ctx.close()
        XX = np.concatenate((np.ones((X.shape[0], 1)), cols.T), axis=1)
        return np.matmul(XX, self.betas_[-last_ind])

    # ---- beyond the reference: every chain's prediction in one pass, and a stable model file
    def predict_all(self, test_data):
        """Predictions of every fitted chain: array (n_chains, n_rows).  Row -i equals predict(test_data, last_ind=i)[:, 0].
        All chains' trees are evaluated in batches on one device context (the reference evaluates one chain per call)."""
        from .device import DeviceContext
        from .tape import flatten
        X = np.ascontiguousarray(np.asarray(test_data, dtype=np.float64))
        K = self.treeNum
        tapes = [flatten(self.roots_[c][k]) for c in range(len(self.roots_)) for k in range(K)]
        out = np.empty((len(self.roots_), X.shape[0]), dtype=np.float64)
        per = max(K, (256 // K) * K)                       # whole chains per launch
        ctx = DeviceContext(X, None, max_batch=per, device=self.device, dtype=self.dtype)
        try:
            for lo in range(0, len(tapes), per):
                cols, _, _ = ctx.eval_tapes(tapes[lo:lo + per])
                for j in range(0, cols.shape[0], K):
                    c = (lo + j) // K
                    XX = np.concatenate((np.ones((X.shape[0], 1)), cols[j:j + K].T), axis=1)
                    out[c] = np.matmul(XX, self.betas_[c])[:, 0]
        finally:
            ctx.close()
        return out

    def save(self, path):
        """Writes the fitted chains (trees as postfix tapes, Beta, RMSE history) and the constructor arguments as JSON.
        The reference offers only pickling of Node graphs; this file does not depend on Python object layout."""
        import json
        from .tape import flatten

        def enc(v):
            v = float(v)
            return v if np.isfinite(v) else ("nan" if np.isnan(v) else ("inf" if v > 0 else "-inf"))
        chains = []
        for c in range(len(self.roots_)):
            trees = []
            for root in self.roots_[c]:
                t = flatten(root)
                trees.append([[int(r["opcode"]), int(r["left"]), int(r["right"]), int(r["feature"]), enc(r["a"]), enc(r["b"])]
                              for r in t])
            chains.append({"trees": trees, "beta": [enc(b) for b in np.asarray(self.betas_[c]).reshape(-1)],
                           "train_err": [enc(e) for e in self.train_err_[c]]})
        doc = {"format": "bsr-hip-model", "version": 1,
               "operators": {"0": "inv", "1": "ln", "2": "neg", "3": "sin", "4": "cos", "5": "exp", "6": "square",
                             "7": "cubic", "8": "+", "9": "*", "10": "terminal", "13": "sub", "14": "div", "15": "log"},
               "node_fields": ["opcode (key of operators)", "left", "right", "feature", "a", "b"],
               "params": {k: getattr(self, k) for k in ("treeNum", "itrNum", "alpha1", "alpha2", "beta", "val")},
               "chains": chains}
        with open(path, "w") as f:
            json.dump(doc, f)

    @classmethod
    def load(cls, path, **kwargs):
        """Rebuilds an estimator written by save(): model(), complexity(), predict(), predict_all() work at once."""
        import json
        from .tape import NODE_DTYPE, unflatten

        def dec(v):
            return float(v)                                # "nan" / "inf" / "-inf" parse as floats
        with open(path) as f:
            doc = json.load(f)
        if doc.get("format") != "bsr-hip-model" or doc.get("version") != 1:
            raise ValueError("not a bsr-hip-model v1 file: %r" % path)
        est = cls(**dict(doc["params"], **kwargs))
        est.roots_, est.betas_, est.train_err_ = [], [], []
        for ch in doc["chains"]:
            roots = []
            for rows in ch["trees"]:
                t = np.zeros(len(rows), dtype=NODE_DTYPE)
                for i, r in enumerate(rows):
                    t[i] = (r[0], r[1], r[2], r[3], dec(r[4]), dec(r[5]))
                roots.append(unflatten(t))
            est.roots_.append(roots)
            est.betas_.append(np.array([dec(b) for b in ch["beta"]], dtype=np.float64).reshape(-1, 1))
            est.train_err_.append([dec(e) for e in ch["train_err"]])
        return est

    # ---- codes/bsr_class.py:77-278
    def fit(self, train_data, train_y):
        self.roots_, self.betas_, self.train_err_ = [], [], []
        y_is_series = hasattr(train_y, "iloc")
        X = np.ascontiguousarray(np.asarray(train_data, dtype=np.float64))
        y = np.ascontiguousarray(np.asarray(train_y, dtype=np.float64).reshape(-1))
        N, d = X.shape
        K = self.treeNum
        if self.devices is not None:
            return self._fit_sharded(X, y, K, y_is_series)
        from .native import default_batch
        batch = int(self.batch) if self.batch else default_batch(N, d, K)
        T = self._table()
        seeds = self.chain_seeds
        n_slots = 1 if seeds is None else max(1, min(self.chains_per_launch, len(seeds), self.itrNum))
        tc, tb = (0, 0)
        if self.engine == "native":
            from .native import batch_shape
            tc, tb = batch_shape(n_slots, batch, K)
        scorer = DeviceScorer(X, y, K, n_chains=n_slots, max_batch=max(4, batch * n_slots), device=self.device,
                              dtype=self.dtype, typical_chains=tc, typical_batch=tb)
        self.stats_ = {"proposals": 0, "accepts": 0, "rank_rejects": 0, "discarded": 0}
        results = []
        try:
            if self.engine == "native":
                results = self._fit_native(scorer, N, d, K, seeds, n_slots, y_is_series)
            elif seeds is None:
                if self.disp:
                    print('starting training...')
                for _ in range(self.itrNum):
                    ch = Chain(0, scorer, N, d, K, beta=self.beta, val=self.val, table=T, y_is_series=y_is_series,
                               feature_range=(X.min(axis=0), X.max(axis=0)))
                    run_chains([ch], scorer, batch_per_chain=batch)
                    rng.set_state(ch.rng_state)      # the next chain continues the same stream
                    results.append(ch.result())
                    results[-1].update(n_rank_rejects=ch.n_rank_rejects, n_discarded=ch.n_discarded)
            else:
                todo = list(range(min(self.itrNum, len(seeds))))
                while todo:
                    wave, todo = todo[:n_slots], todo[n_slots:]
                    chains = []
                    for slot, ci in enumerate(wave):
                        np.random.seed(seeds[ci])
                        chains.append(Chain(slot, scorer, N, d, K, beta=self.beta, val=self.val, table=T,
                                            y_is_series=y_is_series, feature_range=(X.min(axis=0), X.max(axis=0))))
                    run_chains(chains, scorer, batch_per_chain=batch)
                    for ch in chains:
                        results.append(ch.result())
                        results[-1].update(n_rank_rejects=ch.n_rank_rejects, n_discarded=ch.n_discarded)
        finally:
            scorer.close()
        for r in results:
            self.roots_.append(r["roots"])
            self.betas_.append(r["beta"])
            self.train_err_.append(r["errs"])
            self.stats_["proposals"] += r["n_props"]
            self.stats_["accepts"] += r["n_accept"]
            self.stats_["rank_rejects"] += r["n_rank_rejects"]
            self.stats_["discarded"] += r["n_discarded"]
        return

    def _table(self):
        """OpTable of this estimator: Ops / Op_weights / Op_type of codes/bsr_class.py:110-112, or the user's."""
        from .node import OPS, OP_ARITY
        if self.ops is None and self.op_weights is None:
            return P.default_table()
        ops = list(self.ops) if self.ops is not None else list(OPS)
        w = list(self.op_weights) if self.op_weights is not None else [1.0 / len(ops)] * len(ops)
        if len(w) != len(ops):
            raise ValueError("ops and op_weights differ in length")
        return P.OpTable.get(ops, w, [OP_ARITY[o] for o in ops])

    def _fit_sharded(self, X, y, K, y_is_series):
        """One process per GPU of `devices`; this process touches no GPU (codes/bsr_class.py:99, 270-276 sharded)."""
        from .sharded import fit_sharded
        seeds = self.chain_seeds if self.chain_seeds is not None else [1000 + c for c in range(self.itrNum)]
        seeds = [int(v) for v in seeds[:self.itrNum]]
        recs = fit_sharded(X, y, K=K, seeds=seeds, devices=list(self.devices), batch=self.batch or None, val=self.val,
                           beta=self.beta, chains_per_launch=self.chains_per_launch, dtype=self.dtype,
                           y_is_series=y_is_series, ops=None if self.ops is None and self.op_weights is None else self._table().ops,
                           op_weights=None if self.ops is None and self.op_weights is None else self._table().weights)
        self.stats_ = {"proposals": 0, "accepts": 0, "rank_rejects": 0, "discarded": 0}
        for r in recs:
            self.roots_.append(r["roots"])
            self.betas_.append(r["beta"])
            self.train_err_.append(r["errs"])
            self.stats_["proposals"] += r["n_props"]
            self.stats_["accepts"] += r["n_accept"]
            self.stats_["rank_rejects"] += r["n_rank_rejects"]
            self.stats_["discarded"] += r["n_discarded"]
        return

    def _fit_native(self, scorer, N, d, K, seeds, n_slots, y_is_series):
        from .native import default_batch
        batch = int(self.batch) if self.batch else default_batch(N, d, K)
        from .native import NativeEngine
        eng = NativeEngine(scorer.ctx, n_slots, d, beta=self.beta, val=self.val, y_is_series=y_is_series)
        T = self._table()
        eng.set_ops(T.ops, T.weights)
        results = []
        try:
            if seeds is None:
                # chains one after the other on numpy's global stream, exactly like the reference
                eng.set_numpy_state(0)
                for _ in range(self.itrNum):
                    eng.init_chain(0)
                    eng.run(batch_per_chain=batch)
                    results.append(eng.result(0))
                np.random.set_state(eng.get_numpy_state(0))
            else:
                todo = list(range(min(self.itrNum, len(seeds))))
                while todo:
                    wave, todo = todo[:n_slots], todo[n_slots:]
                    for slot, ci in enumerate(wave):
                        eng.seed(slot, seeds[ci])
                        eng.init_chain(slot)
                    eng.run(batch_per_chain=batch)
                    for slot, ci in enumerate(wave):
                        results.append(eng.result(slot))
        finally:
            eng.close()
        return results
