#!/usr/bin/env python3
"""Executed cost of every tape shape in the scoring row pass (run on the GPU box).

    python tools/probes/op_costs.py [--N 100000 --d 10] [--reps 12]            # kernel time per shape (HIP events)
    bash tools/probes/op_costs.sh [--N ... ]                                    # the same under rocprofv3 --pmc

A batch of 64 copies of ONE tape shape is scored `reps` times per shape, shapes in a fixed order; under the counters
the k-th group of `reps` row-pass dispatches belongs to shape k (printed as `shape <k> <name>`), so per-shape VALU /
SALU / LDS instruction counts fall out of the counter file (op_costs.sh does the join).  Differences against the leaf
shape, divided by 64 tapes x row blocks, are the executed instructions per (tape, 128-row block) of an operator."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))

import numpy as np

from bsr.device import DeviceContext
from bsr.node import Node
from bsr.tape import flatten


def leaf(f):
    n = Node(1)
    n.type = 0
    n.feature = np.array([f])
    return n


def un(op, child, a=None, b=None):
    n = Node(0)
    n.type, n.operator, n.left = 1, op, child
    n.a, n.b = a, b
    child.parent = n
    return n


def bi(op, l, r):
    n = Node(0)
    n.type, n.operator, n.left, n.right = 2, op, l, r
    l.parent = r.parent = n
    return n


def shapes(d):
    x = leaf
    s01 = lambda: bi('+', x(0), x(1 % d))
    out = [
        ("leaf", lambda i: x(i % d)),
        ("add_t", lambda i: bi('+', x(i % d), x((i + 1) % d))),
        ("mul_t2", lambda i: bi('*', bi('*', x(i % d), x((i + 1) % d)), x((i + 2) % d))),
        ("ln", lambda i: un('ln', bi('+', x(i % d), x((i + 1) % d)), 1.25, -0.5)),
        ("neg", lambda i: un('neg', bi('+', x(i % d), x((i + 1) % d)))),
        ("square", lambda i: un('square', bi('+', x(i % d), x((i + 1) % d)))),
        ("cubic", lambda i: un('cubic', bi('+', x(i % d), x((i + 1) % d)))),
        ("inv", lambda i: un('inv', bi('+', x(i % d), x((i + 1) % d)))),
        ("sin", lambda i: un('sin', bi('+', x(i % d), x((i + 1) % d)))),
        ("cos", lambda i: un('cos', bi('+', x(i % d), x((i + 1) % d)))),
        ("exp", lambda i: un('exp', bi('+', x(i % d), x((i + 1) % d)))),
        ("sin_leaf(derived)", lambda i: un('sin', x(i % d))),
        ("stack", lambda i: bi('*', bi('+', x(i % d), x((i + 1) % d)), bi('+', x((i + 2) % d), x((i + 3) % d)))),
        ("sin_sin_sin", lambda i: un('sin', un('sin', un('sin', bi('+', x(i % d), x((i + 1) % d)))))),
    ]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=100000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(a.N, a.d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(a.N)
    ctx = DeviceContext(X, y, K=a.K, n_chains=1, max_batch=a.B, dtype=a.dtype)
    cur = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2 % a.d)), un('ln', un('exp', leaf(3 % a.d)), 0.7, -0.2),
           un('cos', leaf(4 % a.d)), un('cubic', leaf(5 % a.d)), bi('+', leaf(6 % a.d), leaf(7 % a.d)),
           un('inv', leaf(8 % a.d)), un('square', leaf(9 % a.d))][:a.K]
    for k, t in enumerate(cur):
        ctx.set_current(0, k, flatten(t))
    ctx.refresh(0)
    info = ctx.info()
    print("geometry", info, flush=True)
    ctx.set_profiling(1)
    chains = np.zeros(a.B, dtype=np.int32)
    ks = (np.arange(a.B) % a.K).astype(np.int32)
    sig = np.full(a.B, 0.8)
    for si, (name, make) in enumerate(shapes(a.d)):
        tapes = [flatten(make(i)) for i in range(a.B)]
        ts = []
        for r in range(a.reps):
            ctx.score_batch(tapes, chains, ks, sig)
            ts.append(ctx.last_timing()[0])
        ts = np.array(ts[2:])
        print("shape %d %s reps %d kernel_us median %.2f min %.2f" % (si, name, a.reps, np.median(ts), ts.min()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
