#!/usr/bin/env python3
"""A streaming context that holds SEVERAL chains (their bases side by side in the chunk buffer: every tape reads its own
chain's) must score a proposal exactly as a context that holds that chain alone: run on the GPU box.

    python tools/probes/stream_multichain_check.py [--N 300077 --d 50 --K 3 --chains 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))

import numpy as np

import stream_check as S
from bsr.device import DeviceContext
from bsr.tape import flatten


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=300077)
    ap.add_argument("--d", type=int, default=50)
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--chains", type=int, default=3)
    a = ap.parse_args()
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(a.N, a.d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(a.N)
    leaf, un, bi = S.leaf, S.un, S.bi
    pool = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2)), un('ln', un('exp', leaf(3)), 0.7, -0.2), un('cos', leaf(4)),
            un('cubic', leaf(5)), bi('+', leaf(6), leaf(7)), un('inv', un('ln', un('square', leaf(8)), 1.0, 1.0)),
            un('square', leaf(9)), un('exp', leaf(10)), bi('*', leaf(11), un('sin', leaf(12))), un('neg', leaf(13)),
            bi('+', un('cos', leaf(14)), leaf(15))]
    B = 60
    trees = S.make_tapes(a.d, B)
    tapes = [flatten(t) for t in trees]
    chains = (np.arange(B) % a.chains).astype(np.int32)
    ks = ((np.arange(B) // a.chains) % a.K).astype(np.int32)
    sig = np.full(B, 0.8)
    multi = DeviceContext(X, y, K=a.K, n_chains=a.chains, max_batch=B)
    print("multi-chain context:", multi.info()["row_pass"], flush=True)
    for c in range(a.chains):
        for k in range(a.K):
            multi.set_current(c, k, flatten(pool[(c * a.K + k) % len(pool)]))
        multi.refresh(c)
    with np.errstate(all="ignore"):
        rm = multi.score_batch(tapes, chains, ks, sig).copy()
    multi.close()
    bad = 0
    for c in range(a.chains):
        one = DeviceContext(X, y, K=a.K, n_chains=1, max_batch=B)
        for k in range(a.K):
            one.set_current(0, k, flatten(pool[(c * a.K + k) % len(pool)]))
        one.refresh(0)
        idx = np.nonzero(chains == c)[0]
        with np.errstate(all="ignore"):
            r1 = one.score_batch([tapes[i] for i in idx], np.zeros(len(idx), dtype=np.int32), ks[idx], sig[idx]).copy()
        one.close()
        for j, i in enumerate(idx):
            same = (rm["rank"][i] == r1["rank"][j]) and (rm["loglik"][i] == r1["loglik"][j] or
                                                         (np.isnan(rm["loglik"][i]) and np.isnan(r1["loglik"][j])))
            if not same:
                bad += 1
                print("  chain %d proposal %d: %r %r vs alone %r %r" % (c, i, rm["rank"][i], rm["loglik"][i], r1["rank"][j],
                                                                      r1["loglik"][j]), flush=True)
    print("MULTI-CHAIN STREAM CHECK", "OK" if bad == 0 else "FAILED (%d)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
