#!/usr/bin/env python3
"""Staged check of the streaming row pass (bsr_stream.hip) on the GPU box, loud about where it stands:

    python tools/probes/stream_check.py [--N 300000 --d 50 --K 3]

Scores a batch of 64 mixed tapes (chains of every operator, a stack tape, a 20-entry chain, ln-heavy tapes) through a
context whose slices stream through LDS, and checks every log-likelihood against the CPU oracle (vectorised flavour);
then the same batch one tape at a time and in reverse order: every score must repeat bit for bit."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np

import bsr_oracle as O
from bsr.device import DeviceContext
from bsr.node import Node
from bsr.tape import flatten


def leaf(f):
    n = Node(1)
    n.type = 0
    n.feature = np.array([f])
    return n


def un(op, c, a=None, b=None):
    n = Node(0)
    n.type, n.operator, n.left, n.a, n.b = 1, op, c, a, b
    c.parent = n
    return n


def bi(op, l, r):
    n = Node(0)
    n.type, n.operator, n.left, n.right = 2, op, l, r
    l.parent = r.parent = n
    return n


def ocopy(n):
    m = O.ONode(n.depth)
    m.type, m.operator, m.feature, m.a, m.b = n.type, n.operator, n.feature, n.a, n.b
    m.left = ocopy(n.left) if n.left is not None else None
    m.right = ocopy(n.right) if n.right is not None else None
    return m


def make_tapes(d, B):
    x = leaf
    rs = np.random.RandomState(3)
    out = []
    ops1 = ['inv', 'neg', 'sin', 'cos', 'exp', 'square', 'cubic']
    for i in range(B):
        f = [int(v) for v in rs.randint(0, d, size=6)]
        kind = i % 12
        if kind == 0:
            t = x(f[0])
        elif kind == 1:
            t = bi('+', x(f[0]), x(f[1]))
        elif kind == 2:
            t = bi('*', bi('*', x(f[0]), x(f[1])), x(f[2]))
        elif kind == 3:
            t = un('ln', bi('+', x(f[0]), x(f[1])), 1.25, -0.5)
        elif kind == 4:
            t = un(ops1[i % 7], bi('+', x(f[0]), x(f[1])))
        elif kind == 5:
            t = un(ops1[(i + 3) % 7], x(f[0]))                       # derived-column candidate
        elif kind == 6:
            t = bi('*', bi('+', x(f[0]), x(f[1])), bi('+', x(f[2]), x(f[3])))   # needs the stack
        elif kind == 7:
            t = x(f[0])
            for j in range(19):                                       # a chain of 20 entries: beyond the scalar registers
                t = bi('+', t, x(f[j % 6])) if j % 2 else un('neg', t)
        elif kind == 8:
            t = un('ln', un('ln', un('ln', un('ln', x(f[0]), 1.1, 0.1), 0.9, -0.1), 1.2, 0.3), 0.8, 0.2)   # four ln nodes
        elif kind == 9:
            t = un('sin', un('ln', bi('*', x(f[0]), un('cos', x(f[1]))), 0.7, 0.3))
        elif kind == 10:
            t = un('exp', un('neg', un('square', bi('+', x(f[0]), x(f[1])))))
        else:
            t = bi('+', un('inv', bi('+', x(f[0]), un('ln', x(f[1]), 1.0, 4.0))), x(f[2]))
        out.append(t)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=300000)
    ap.add_argument("--d", type=int, default=50)
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--B", type=int, default=64)
    a = ap.parse_args()
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(a.N, a.d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(a.N)
    print("creating context", flush=True)
    ctx = DeviceContext(X, y, K=a.K, n_chains=1, max_batch=a.B)
    print("geometry", ctx.info(), flush=True)
    pool = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2 % a.d)), un('ln', un('exp', leaf(3 % a.d)), 0.7, -0.2),
            un('cos', leaf(4 % a.d)), un('cubic', leaf(5 % a.d)), bi('+', leaf(6 % a.d), leaf(7 % a.d)),
            un('inv', un('ln', un('square', leaf(8 % a.d)), 1.0, 1.0)), un('square', leaf(9 % a.d))]
    cur = pool[:a.K]
    for k, t in enumerate(cur):
        ctx.set_current(0, k, flatten(t))
    ctx.refresh(0)
    print("chain set", flush=True)
    trees = make_tapes(a.d, a.B)
    tapes = [flatten(t) for t in trees]
    chains = np.zeros(a.B, dtype=np.int32)
    ks = (np.arange(a.B) % a.K).astype(np.int32)
    sig = np.full(a.B, 0.8)
    print("scoring one leaf", flush=True)
    r1 = ctx.score_batch(tapes[:1], chains[:1], ks[:1], sig[:1])
    print("  ->", r1["loglik"][0], int(r1["rank"][0]), flush=True)
    print("scoring the batch", flush=True)
    res = ctx.score_batch(tapes, chains, ks, sig).copy()
    print("  done; checking against the oracle", flush=True)
    import pandas as pd
    df = pd.DataFrame(X)
    with np.errstate(all="ignore"):
        cols = np.stack([O.allcal(ocopy(t), df)[:, 0] for t in cur], axis=1)
        bad = 0
        worst = 0.0
        for i, t in enumerate(trees):
            z = O.allcal(ocopy(t), df)[:, 0]
            want = O.score_proposal(cols, int(ks[i]), z, y, 0.8)
            got_rank, got_ll = int(res["rank"][i]), float(res["loglik"][i])
            if want["rank"] != got_rank:
                bad += 1
                print("  tape %d (kind %d): rank %d, oracle %d" % (i, i % 12, got_rank, want["rank"]), flush=True)
                continue
            if want["rank"] < a.K:
                continue
            rel = abs(got_ll - want["loglik"]) / abs(want["loglik"])
            worst = max(worst, rel)
            if not rel < 1e-6:
                bad += 1
                print("  tape %d (kind %d): loglik %.12g, oracle %.12g (rel %.2e)" % (i, i % 12, got_ll, want["loglik"], rel), flush=True)
    print("oracle check: %d bad of %d, worst relative difference %.2e" % (bad, a.B, worst), flush=True)
    # bit-equality: alone, reversed
    diff = 0
    rev = ctx.score_batch(tapes[::-1], chains, ks[::-1].copy(), sig).copy()[::-1]
    for f in ("loglik", "sse", "rank"):
        same = (res[f] == rev[f]) | ((res[f] != res[f]) & (rev[f] != rev[f]))
        diff += int((~same).sum())
    for i in range(0, a.B, 7):
        one = ctx.score_batch(tapes[i:i + 1], chains[:1], ks[i:i + 1], sig[:1])
        if not (one["loglik"][0] == res["loglik"][i] or (one["loglik"][0] != one["loglik"][0] and res["loglik"][i] != res["loglik"][i])):
            diff += 1
            print("  tape %d alone: %.17g, in the batch %.17g" % (i, one["loglik"][0], res["loglik"][i]), flush=True)
    print("bit-equality (reversed batch, single tapes): %d differences" % diff, flush=True)
    ctx.set_profiling(1)
    ts = []
    for r in range(12):
        ctx.score_batch(tapes, chains, ks, sig)
        ts.append(ctx.last_timing()[0])
    print("row pass %.1f us (median of 12)" % float(np.median(ts[2:])), flush=True)
    ctx.close()
    print("STREAM CHECK", "OK" if bad == 0 and diff == 0 else "FAILED", flush=True)
    return 0 if bad == 0 and diff == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
