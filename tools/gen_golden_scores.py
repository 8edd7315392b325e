#!/usr/bin/env python3
"""Fixture g9: the ORACLE's scores for the random batches of tests/test_gpu_kernels.py::test_score_batch_vs_oracle,
computed once in the build container (CPU only, no reference import needed: the oracle is pinned on the reference's own
fixtures by tests/test_oracle_golden.py), so that the GPU test's 1e-6 end-to-end gate does not depend on the numpy build
of the box it runs on (numpy's SIMD exp / power / sin dispatch differs from host to host in the last digits).

Per case and proposal: rank, loglik, scale, and `chaotic` -- True when the oracle's own log-likelihood moves by more than
1e-7 relative under a one-ulp relative perturbation of X: the only proposals the GPU test may exempt from the 1e-6 bound.

    python tools/gen_golden_scores.py            # writes tests/golden/g9_score_batch.json
"""
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bsr_oracle as O

CASES = [(3000, 6, 3, 48, 1), (1000, 4, 8, 40, 2), (777, 3, 1, 24, 3), (5000, 10, 2, 64, 4), (130, 2, 3, 64, 5),
         (800, 4, 4, 32, 6), (800, 5, 5, 32, 7), (600, 4, 6, 32, 8), (600, 5, 7, 32, 9)]


def rand_trees(rs_seed, n, d, min_nodes=1):
    out = []
    np.random.seed(rs_seed)
    while len(out) < n:
        root = O.ONode(0)
        O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
        if O.count_nodes(root) >= min_nodes and O.count_nodes(root) < 200:
            out.append(root)
    return out


def enc(v):
    v = float(v)
    return v if np.isfinite(v) else ("nan" if np.isnan(v) else ("inf" if v > 0 else "-inf"))


def main():
    out = {"_generated_by": "tools/gen_golden_scores.py", "numpy": np.__version__, "cases": {}}
    for N, d, K, B, seed in CASES:
        rs = np.random.RandomState(seed)
        X = rs.uniform(-3, 3, size=(N, d))
        y = 1.35 * X[:, 0] * X[:, 1 % d] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1 % d] - 1)) + 0.1 * rs.standard_normal(N)
        trees = rand_trees(100 + seed, 2 * K + B, d)
        cur = [trees[:K], trees[K:2 * K]]
        cands = trees[2 * K:]
        Xdf = pd.DataFrame(X)
        with np.errstate(all="ignore"):
            cur_cols = [np.stack([O.allcal(t, Xdf)[:, 0] for t in cur[c]], axis=1) for c in range(2)]
        chains = rs.randint(0, 2, size=B)
        ks = rs.randint(0, K, size=B)
        sig = rs.uniform(0.3, 3.0, size=B)
        rec = {"rank": [], "loglik": [], "scale": [], "chaotic": [], "express": []}
        for i in range(B):
            with np.errstate(all="ignore"):
                col = O.allcal(cands[i], Xdf)[:, 0]
            want = O.score_proposal(cur_cols[chains[i]], ks[i], col, y, sig[i])
            chaotic = False
            if want["rank"] == K:
                vals = []
                for eps in (2.0 ** -52, -2.0 ** -52, 2.0 ** -51):
                    with np.errstate(all="ignore"):
                        colp = O.allcal(cands[i], pd.DataFrame(X * (1.0 + eps)))[:, 0]
                    vals.append(O.score_proposal(cur_cols[chains[i]], ks[i], colp, y, sig[i]).get("loglik", np.nan))
                spread = max(abs(v - want["loglik"]) for v in vals)
                chaotic = bool(spread > 1e-7 * abs(want["loglik"]))
            rec["rank"].append(int(want["rank"]))
            rec["loglik"].append(enc(want.get("loglik", np.nan)))
            rec["scale"].append(enc(want.get("scale", np.nan)))
            rec["chaotic"].append(chaotic)
            rec["express"].append(O.express(cands[i]))
        key = "N=%d d=%d K=%d B=%d seed=%d" % (N, d, K, B, seed)
        out["cases"][key] = rec
        print(key, "full rank", sum(r == K for r in rec["rank"]), "chaotic", sum(rec["chaotic"]))
    with open(os.path.join(ROOT, "tests", "golden", "g9_score_batch.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
