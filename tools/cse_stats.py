#!/usr/bin/env python3
"""How much arithmetic a cross-tape common-subexpression pass could save on the real move mix (CPU only; tool, uses the
oracle as the scorer -- the same batch generation as tools/shape_stats.py).

Every batch is B speculative proposals of ONE chain state: most candidates are the current tree with one subtree
changed, so subtrees repeat across the batch's tapes.  Per batch, with the interpreter's cost of a node (the staging
cost model of csrc/bsr_stage.hip: stage_tapes, in half-instructions per row pair): total cost of all nodes; cost of the
DISTINCT subtrees (each computed once per batch -- the upper bound of any CSE: it ignores the column an intermediate
would have to be stored in and read from); the part of the difference that derived columns (`terminal, unary op`
pairs, admitted up to 8 per batch) already take.

    python tools/cse_stats.py [--K 3] [--batches 64] [--seeds 4]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mcmc-symreg_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np

# cost of one node in the interpreter (half vector instructions per row pair; csrc/bsr_stage.hip: stage_tapes)
COST = {"inv": 35, "ln": 5, "neg": 3, "sin": 77, "cos": 77, "exp": 59, "square": 3, "cubic": 23, "+": 6, "*": 6,
        "sub": 6, "div": 40, "log": 90}
LOAD = 2       # a terminal: one LDS read
COLUMN = 4     # what reading a stored intermediate back costs (LDS read + the store amortised) -- charged per use


def key_and_cost(node, acc):
    """structural key of the subtree (operands of + and * order-free) and its cost; acc collects (key, cost, size)"""
    if node.type == 0:
        k = ("x", int(np.asarray(node.feature).reshape(-1)[0]))
        c, n = LOAD, 1
    elif node.type == 1:
        lk, lc, ln_ = key_and_cost(node.left, acc)
        par = (float(node.a), float(node.b)) if node.operator == "ln" else ()
        k = (node.operator, par, lk)
        c, n = lc + COST.get(node.operator, 10), ln_ + 1
    else:
        lk, lc, ln_ = key_and_cost(node.left, acc)
        rk, rc, rn = key_and_cost(node.right, acc)
        if node.operator in ("+", "*") and repr(rk) < repr(lk):
            lk, rk = rk, lk
        k = (node.operator, lk, rk)
        c, n = lc + rc + COST.get(node.operator, 6), ln_ + rn + 1
    acc.append((k, c, n, node.type))
    return k, c, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--N", type=int, default=4000)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--batches", type=int, default=64)
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--burnin", type=int, default=300)
    a = ap.parse_args()
    from bench import synth
    from bsr.chain import Chain, run_chains
    from test_host_driver import OracleScorer
    tot = dist = derived_saved = 0.0
    nodes = dnodes = 0
    n_b = 0
    top = collections.Counter()
    for sd in range(a.seeds):
        X, y = synth(a.N, a.d, seed=0)
        sc = OracleScorer(X, y, a.K, n_chains=1, max_batch=a.B)
        np.random.seed(1000 + sd)
        ch = Chain(0, sc, a.N, a.d, a.K, val=10 ** 9)
        run_chains([ch], sc, batch_per_chain=a.B, max_props=a.burnin)
        for _ in range(a.batches):
            cands = ch.generate(a.B)
            seen = {}
            uses = collections.Counter()
            b_tot = 0
            for cd in cands:
                acc = []
                k, c, n = key_and_cost(cd.root, acc)
                b_tot += c
                nodes += n
                for (kk, cc, nn, ty) in acc:
                    own = cc - sum(0 for _ in ())   # cost of the whole subtree; the node's own cost is added below
                    uses[kk] += 1
                    seen[kk] = (cc, nn, ty)
            # distinct subtrees: every distinct key pays its ROOT node once (its children are keys of their own)
            b_dist = 0
            for kk, (cc, nn, ty) in seen.items():
                if ty == 0:
                    own = LOAD
                elif ty == 1:
                    own = cc - seen[kk[2]][0]
                else:
                    own = cc - seen[kk[1]][0] - seen[kk[2]][0]
                # a shared intermediate is stored once and read back by its other uses
                b_dist += own + (COLUMN * (uses[kk] - 1) if (ty != 0 and uses[kk] > 1) else 0)
                dnodes += 1
                if ty != 0 and uses[kk] > 1:
                    top[(kk[0], nn)] += uses[kk] - 1
            # what derived columns already take: `terminal, unary` pairs used at least twice, best 8 by saving
            dcand = []
            for kk, (cc, nn, ty) in seen.items():
                if ty == 1 and kk[0] != "ln" and kk[2][0] == "x" and uses[kk] > 1:
                    dcand.append((COST.get(kk[0], 10) * (uses[kk] - 1)))
            derived_saved += sum(sorted(dcand, reverse=True)[:8])
            tot += b_tot
            dist += b_dist
            n_b += 1
            ch.rng_state = ch._end_state
    print("K=%d, %d batches of %d: %.1f nodes per tape, %.1f distinct subtrees per batch (of %.1f nodes)" %
          (a.K, n_b, a.B, nodes / (n_b * a.B), dnodes / n_b, nodes / n_b))
    print("interpreter cost per batch (half-instructions per row pair): all nodes %.0f, distinct subtrees once %.0f "
          "(-%.1f %%); derived columns already save %.0f (%.1f %%); left for a cross-tape pass: %.1f %%" %
          (tot / n_b, dist / n_b, 100 * (1 - dist / tot), derived_saved / n_b, 100 * derived_saved / tot,
           100 * (1 - dist / tot) - 100 * derived_saved / tot))
    print("most repeated shared subtrees (root op, nodes): extra uses over all batches")
    for (op, nn), c in top.most_common(10):
        print("   %-7s %2d nodes: %d" % (op, nn, c))


if __name__ == "__main__":
    main()
