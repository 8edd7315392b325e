"""Lifetime and ordering hazards of the first batch of a fresh context (the withdrawn fused tail kernel "read
a zero chain block in the first batch after a refresh, in alternate fresh contexts" -- the signature of a stream-order or
memory-reuse hazard; the shipped path must be immune by test, not by being slow enough to hide it).

200 contexts are created and destroyed one after the other -- the allocator hands the next one the blocks the last one
freed -- on an ill-conditioned K = 8 chain state whose scores move with the last bit of anything that is off.  Every
context sets its trees, refreshes and scores ITS FIRST batch at once (nothing in between that would give a straggling
fill or copy time to land); every one of the 200 results must equal the first byte for byte.  Half of the contexts run
with BSR_POISON=1 semantics exercised by tests/test_gpu_edges.py as well: here the interleaving of poisoned and clean
allocations is the point (a buffer that was clean in one life and is read before it is written in the next)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bsr_oracle as O
from bsr.device import DeviceContext
from bsr.node import Node
from bsr.tape import flatten
from conftest import node_from_spec, spec_from_node


def _workload(K, N=20000, d=6, B=64):
    rs = np.random.RandomState(23)
    X = rs.uniform(-3, 3, size=(N, d))
    y = X[:, 0] * X[:, 1] + np.sin(X[:, 2]) + 0.1 * rs.standard_normal(N)
    np.random.seed(31)
    trees = []
    while len(trees) < K + B:
        root = O.ONode(0)
        O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
        if O.count_nodes(root) < 30:
            trees.append(node_from_spec(spec_from_node(root)))
    tapes = [flatten(t) for t in trees[K:]]
    ks = (np.arange(B) % K).astype(np.int32)
    for j in range(min(K, 4)):           # repeats of current trees: the span shortcut and the residual route both run
        tapes[j] = flatten(trees[j])
        ks[j] = j
    neg = Node(0)
    neg.type, neg.operator, neg.left = 1, 'neg', node_from_spec(spec_from_node(trees[0]))
    neg.left.parent = neg
    tapes[5] = flatten(neg)
    ks[5] = 0
    sig = rs.uniform(0.5, 2.0, size=B)
    return X, y, [flatten(t) for t in trees[:K]], tapes, ks, sig


@pytest.mark.parametrize("K", [3, 8])
def test_two_hundred_fresh_contexts_score_their_first_batch_identically(K, monkeypatch):
    X, y, cur, tapes, ks, sig = _workload(K)
    B = len(tapes)
    zeros = np.zeros(B, np.int32)

    def first_batch(poison, async_submit):
        if poison:
            monkeypatch.setenv("BSR_POISON", "1")
        else:
            monkeypatch.delenv("BSR_POISON", raising=False)
        c = DeviceContext(X, y, K=K, n_chains=1, max_batch=B)
        try:
            for k in range(K):
                c.set_current(0, k, cur[k])
            c.refresh(0)
            if async_submit:             # the non-blocking route, two batches queued back to back on fresh slots
                from bsr.tape import pack
                rows, off = pack(tapes)
                t0 = c.score_submit(rows, off, zeros, ks, sig)
                t1 = c.score_submit(rows, off, zeros, ks, sig)
                from bsr import _lib
                o0 = np.zeros(B, dtype=_lib.SCORE_DTYPE)
                o1 = np.zeros(B, dtype=_lib.SCORE_DTYPE)
                c.score_wait(t0, o0)
                c.score_wait(t1, o1)
                assert o0.tobytes() == o1.tobytes()
                return o0
            return c.score_batch(tapes, zeros, ks, sig).copy()
        finally:
            c.close()

    base = first_batch(False, False)
    assert (base["rank"] >= 0).all() and np.isfinite(base["loglik"][base["rank"] == K]).all()
    bad = []
    for i in range(200):
        got = first_batch(poison=(i % 4) in (1, 2), async_submit=(i % 3) == 0)
        if got.tobytes() != base.tobytes():
            d = np.nonzero(got["loglik"].view(np.uint64) != base["loglik"].view(np.uint64))[0]
            bad.append((i, len(d)))
    assert not bad, "contexts whose first batch differs (index, differing log-likelihoods): %r" % bad[:10]
