#!/usr/bin/env python3
"""Would an SSE memo pay?  (round 4's review, item 7: "first MEASURE the hit rate of an SSE memo keyed by (identity hash incl.
ln params, k, chain-state version) ... build it only if the measured hit rate is >= 20 %".)

SSE does not depend on sigma (codes/funcs.py:1162-1173), so a chain that proposes the same tree for the same slot k twice
between two accepts could take the second score from a table.  This probe runs the Python sampler (bsr.chain, the
reference's draw order) on a CPU scorer -- which candidates a chain proposes does not depend on the scorer's speed -- and
counts, over the CONSUMED proposals of a chain, how many repeat a (tape bytes, k) pair already scored since the chain's
last accept.  Tape bytes include the ln parameters, which are drawn afresh for every proposal that touches an ln node,
so a repeat is an exact repeat.

    python tools/memo_probe.py [--K 3] [--d 10] [--props 4000] [--seed 1000]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--N", type=int, default=300)
    ap.add_argument("--props", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    from test_host_driver import OracleScorer
    from bsr.chain import Chain, run_chains
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(a.N, a.d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(a.N)
    scorer = OracleScorer(X, y, a.K, max_batch=a.batch)
    seen, stats = {}, {"consumed": 0, "hits": 0, "hits_structure_only": 0, "accepts": 0, "scored": 0}
    seen_struct = {}
    orig_score = scorer.score

    def score(tapes, chains, ks, sigmas):
        stats["scored"] += len(tapes)
        score.last = [(t.tobytes(), int(k), t[["opcode", "feature"]].tobytes()) for t, k in zip(tapes, ks)]
        return orig_score(tapes, chains, ks, sigmas)
    scorer.score = score
    np.random.seed(a.seed)
    ch = Chain(0, scorer, a.N, a.d, a.K, val=10 ** 9)
    orig_consume = ch.consume

    def consume(res, idx):
        n0, a0 = ch.n_props, ch.n_accept
        orig_consume(res, idx)
        used = ch.n_props - n0
        for j in range(used):
            key = score.last[list(idx)[j]]
            stats["consumed"] += 1
            if (key[0], key[1]) in seen:
                stats["hits"] += 1
            if (key[2], key[1]) in seen_struct:
                stats["hits_structure_only"] += 1
            seen[(key[0], key[1])] = 1
            seen_struct[(key[2], key[1])] = 1
        if ch.n_accept != a0:
            stats["accepts"] += ch.n_accept - a0
            seen.clear()
            seen_struct.clear()
    ch.consume = consume
    run_chains([ch], scorer, batch_per_chain=a.batch, max_props=a.props)
    c = max(1, stats["consumed"])
    print("K=%d d=%d seed=%d: %d consumed proposals (%d scored), %d accepts" % (a.K, a.d, a.seed, stats["consumed"], stats["scored"], stats["accepts"]))
    print("  exact repeats of a (tape incl. ln parameters, k) since the chain's last accept: %d = %.1f %% of the consumed proposals"
          % (stats["hits"], 100.0 * stats["hits"] / c))
    print("  repeats of the same STRUCTURE (opcodes and features; ln parameters may differ -- not a memo hit): %d = %.1f %%"
          % (stats["hits_structure_only"], 100.0 * stats["hits_structure_only"] / c))


if __name__ == "__main__":
    main()
