#!/usr/bin/env python3
"""Pass-1 cost structure on synthetic tapes (run on the GPU box): fixed per-sweep cost vs per-node vs transcendental."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
import numpy as np
from bsr.device import DeviceContext
from bsr.node import Node
from bsr.tape import flatten

def leaf(f):
    n = Node(1); n.type = 0; n.feature = np.array([f]); return n
def un(op, c, a=None, b=None):
    n = Node(0); n.type, n.operator, n.left, n.a, n.b = 1, op, c, a, b; c.parent = n; return n
def bi(op, l, r):
    n = Node(0); n.type, n.operator, n.left, n.right = 2, op, l, r; l.parent = r.parent = n; return n

N, d, K, B = int(os.environ.get("N", 100000)), 10, int(os.environ.get("K", 3)), int(os.environ.get("B", 64))
rs = np.random.RandomState(0)
X = rs.uniform(-3, 3, size=(N, d)); y = rs.standard_normal(N)
ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=B)
cur = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2)), un('ln', leaf(3), 0.7, -0.2), leaf(4), un('cos', leaf(5)),
       un('square', leaf(6)), bi('+', leaf(7), leaf(8)), un('exp', leaf(9))]
for k in range(K):
    ctx.set_current(0, k, flatten(cur[k]))
ctx.refresh(0)
ctx.set_profiling(2)
def chain_of(op, n):
    t = leaf(1)
    for _ in range(n):
        t = un(op, t, 0.9, 0.1) if op == 'ln' else un(op, t)
    return t
cases = {
    "x1": leaf(1),
    "x1*x2": bi('*', leaf(1), leaf(2)),
    "(x1*x2)+(x3*x4)": bi('+', bi('*', leaf(1), leaf(2)), bi('*', leaf(3), leaf(4))),
    "neg^4(x1)": chain_of('neg', 4),
    "ln^4(x1)": chain_of('ln', 4),
    "ln^16(x1)": chain_of('ln', 16),
    "sin(x1)": chain_of('sin', 1),
    "sin^4(x1)": chain_of('sin', 4),
    "exp(x1)": chain_of('exp', 1),
    "inv(x1)": chain_of('inv', 1),
    "cubic(x1)": chain_of('cubic', 1),
}
only = os.environ.get("CASE")
for name, tree in cases.items():
    if only and name != only:
        continue
    tapes = [flatten(tree)] * B
    chains = np.zeros(B, np.int32); ks = (np.arange(B) % K).astype(np.int32); sig = np.ones(B)
    us = np.zeros(5)
    for it in range(25):
        ctx.score_batch(tapes, chains, ks, sig)
        if it >= 5: us += ctx.last_timing()
    us /= 20
    print("%-18s pass1 %7.1f us  solve %5.1f  total %6.1f   per proposal-row %.3f ns" % (name, us[0], us[1], us[4], us[0] * 1e3 / (B * N)))
