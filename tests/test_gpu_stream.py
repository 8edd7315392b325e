"""The streaming row pass (csrc/bsr_stream.hip: k_stream) against the CPU oracle on a context whose slices stream through
LDS: a batch of 64 tapes built to reach every path of the kernel -- chains of every operator (the scalar-register
interpreter), a pushed operand (its one-entry register stack), a 20-entry chain and a four-ln tape (the general stack
machine behind it), derived-column candidates, repeats of current trees (span shortcut / residual route) -- at K = 3
(four sets of sums per wave) and K = 8 (two), N not a multiple of 128 (the block that holds row N is a leftover unit).
Log-likelihoods within 1e-6 relative (codes/funcs.py:1147-1174), rank decisions exact (codes/funcs.py:1226); and every
score must repeat bit for bit whatever else shares its launch (alone, reversed batch)."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))

import bsr_oracle as O
from bsr.device import DeviceContext
from bsr.tape import flatten


@pytest.mark.parametrize("N,d,K", [(300_077, 50, 3), (300_077, 24, 8)])
def test_streaming_pass_against_the_oracle_and_itself(N, d, K):
    import stream_check as S
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    B = 64
    ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=B)
    try:
        info = ctx.info()
        assert info["row_pass"] == "k_stream", info
        leaf, un, bi = S.leaf, S.un, S.bi
        pool = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2)), un('ln', un('exp', leaf(3)), 0.7, -0.2), un('cos', leaf(4)),
                un('cubic', leaf(5)), bi('+', leaf(6), leaf(7)), un('inv', un('ln', un('square', leaf(8)), 1.0, 1.0)),
                un('square', leaf(9))]
        cur = pool[:K]
        for k, t in enumerate(cur):
            ctx.set_current(0, k, flatten(t))
        ctx.refresh(0)
        trees = S.make_tapes(d, B)
        trees[3] = un('sin', leaf(2)) if K > 1 else trees[3]         # tree 1 again (k = 3 % K below may differ: still in the span)
        tapes = [flatten(t) for t in trees]
        chains = np.zeros(B, dtype=np.int32)
        ks = (np.arange(B) % K).astype(np.int32)
        sig = np.full(B, 0.8)
        res = ctx.score_batch(tapes, chains, ks, sig).copy()
        df = pd.DataFrame(X)
        with np.errstate(all="ignore"):
            cols = np.stack([O.allcal(S.ocopy(t), df)[:, 0] for t in cur], axis=1)
            for i, t in enumerate(trees):
                want = O.score_proposal(cols, int(ks[i]), O.allcal(S.ocopy(t), df)[:, 0], y, 0.8)
                assert int(res["rank"][i]) == want["rank"], (i, i % 12, int(res["rank"][i]), want["rank"])
                if want["rank"] == K:
                    rel = abs(float(res["loglik"][i]) - want["loglik"]) / abs(want["loglik"])
                    assert rel < 1e-6, (i, i % 12, float(res["loglik"][i]), want["loglik"])
        rev = ctx.score_batch(tapes[::-1], chains, ks[::-1].copy(), sig).copy()[::-1]
        assert rev.tobytes() == res.tobytes()
        for i in range(0, B, 5):
            one = ctx.score_batch(tapes[i:i + 1], chains[:1], ks[i:i + 1], sig[:1])
            assert one.tobytes() == res[i:i + 1].tobytes(), i
    finally:
        ctx.close()
