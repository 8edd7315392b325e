"""Which row pass -- and which geometry -- every BASELINE configuration selects.  The parity tests compare values; a
context that silently fell back to the work-queue pass (k_rows) would pass all of them and cost 2-3x.  bsr_ctx_info
exposes what the context decided (one decision per context, for its life: a proposal's partial sums must not depend on
the batch); this test pins it, and scores a batch to make sure the chosen kernel is the one that runs (the profiling
timer of the row pass only ticks for the pass that was launched)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

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


# (name, N, d, K, chains, max_batch, dtype) -> expected (row_pass, tape_groups, row_slices, blocks_per_slice)
CASES = [
    ("C2", 100_000, 10, 3, 1, 64, "f64", ("k_tile1a", 1, 98, 7)),
    ("C3", 100_000, 10, 8, 1, 64, "f64", ("k_tile1", 1, 192, 4)),
    ("C4 share", 100_000, 10, 3, 8, 256, "f64", ("k_tile1a", 4, 98, 7)),
    ("C5", 1_000_000, 50, 3, 1, 64, "f64", ("k_stream", 1, 256, 30)),
    ("C5 f32", 1_000_000, 50, 3, 1, 64, "f32", ("k_stream", 1, 256, 30)),
]


@pytest.mark.parametrize("name,N,d,K,chains,max_batch,dtype,want", CASES, ids=[c[0] for c in CASES])
def test_the_regime_each_baseline_config_selects(name, N, d, K, chains, max_batch, dtype, want):
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    ctx = DeviceContext(X, y, K=K, n_chains=chains, max_batch=max_batch, dtype=dtype)
    try:
        info = ctx.info()
        assert info["row_pass"] == want[0], (name, info)
        for key, w in zip(("tape_groups", "row_slices", "blocks_per_slice"), want[1:]):
            if w is not None:
                assert info[key] == w, (name, key, info)
        assert info["streaming"] == (want[0] == "k_stream") and info["slices_whole"] == (want[0] in ("k_tile1", "k_tile1a"))
        # ... and the kernel runs: one batch through it
        pool = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2)), un('ln', un('exp', leaf(3)), 0.7, -0.2), un('cos', leaf(4)),
                un('cubic', leaf(5)), bi('+', leaf(6), leaf(7)), un('inv', un('ln', un('square', leaf(8)), 1.0, 1.0)),
                un('square', leaf(9))]
        for c in range(chains):
            for k in range(K):
                ctx.set_current(c, k, flatten(pool[k]))
            ctx.refresh(c)
        B = min(max_batch, 32)
        tapes = [flatten(bi('+', un('sin', leaf(i % d)), leaf((i + 1) % d))) for i in range(B)]
        ctx.set_profiling(1)
        res = ctx.score_batch(tapes, np.arange(B, dtype=np.int32) % chains, np.arange(B, dtype=np.int32) % K, np.full(B, 0.8))
        assert (res["rank"] == K).all() and np.isfinite(res["loglik"]).all()
        assert ctx.last_timing()[0] > 0.0
    finally:
        ctx.close()


@pytest.mark.parametrize("K", [3, 6])
def test_wildly_scaled_columns_leave_no_nan_in_the_rotations(K):
    """Column norms up to ~1e270 apart: the Jacobi rotations' reciprocal / reciprocal-square-root estimates (csrc/bsr_solve.h:
    rot_coeffs) must fall back to the identity rotation where their Newton steps would make NaN of an overflow -- a NaN
    there spreads into every score of the proposal.  The rank decision is the reference's (matrix_rank, codes/funcs.py:1226)."""
    import pandas as pd
    import bsr_oracle as O
    from conftest import spec_from_node
    rs = np.random.RandomState(3)
    N, d = 4000, 6
    X = rs.uniform(0.5, 3, size=(N, d))
    y = X[:, 0] + X[:, 1] * X[:, 2] + 0.1 * rs.standard_normal(N)
    scales = [1e135, 1e-135, 1.0, 1e60, 1e-60, 1e100][:K]
    cur = [un('ln', leaf(k % d), a, 0.0) for k, a in enumerate(scales)]
    cands = [leaf(3), un('ln', leaf(4), 1e-130, 0.0), un('ln', leaf(5), 1e130, 0.0), bi('*', leaf(0), leaf(1))]
    ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=16)
    try:
        for k in range(K):
            ctx.set_current(0, k, flatten(cur[k]))
        ctx.refresh(0)
        tapes, ks = [], []
        for cd in cands:
            for k in range(K):
                tapes.append(flatten(cd))
                ks.append(k)
        B = min(16, len(tapes))
        res = ctx.score_batch(tapes[:B], np.zeros(B, np.int32), np.array(ks[:B], np.int32), np.full(B, 0.9))
        df = pd.DataFrame(X)
        with np.errstate(all="ignore"):
            cols = [O.allcal(O.tree_from_json(spec_from_node(t)), df)[:, 0] for t in cur]
            for i in range(B):
                out = np.stack(cols, axis=1).copy()
                out[:, ks[i]] = O.allcal(O.tree_from_json(spec_from_node(cands[i // K])), df)[:, 0]
                want_rank = int(np.linalg.matrix_rank(out))
                assert (res["rank"][i] == K) == (want_rank == K), (i, res["rank"][i], want_rank)
                if res["rank"][i] == K:
                    assert np.isfinite(res["loglik"][i]) and np.isfinite(res["beta"][i][:K]).all(), (i, res[i])
                assert not np.isnan(res["smax"][i]) and not np.isnan(res["smin"][i]), (i, res[i])
    finally:
        ctx.close()


def test_a_context_tuned_for_its_callers_batches_keeps_every_batch_on_the_same_sums():
    """bsr_ctx_create_tuned: eight chains, 256 proposals at most, but batches of two chains and 64 proposals as a rule (the
    native sampler's chain groups).  The geometry is the single chain's (98 long slices, the assembly tape loop) instead
    of 256 slices of three blocks; a batch of all eight chains still scores -- through the chunked kernel, its columns
    no longer fit LDS whole -- and every proposal gets the bytes it gets in a narrow batch, and the oracle's value."""
    import pandas as pd
    import bsr_oracle as O
    from conftest import spec_from_node
    rs = np.random.RandomState(4)
    N, d, K, C = 100_000, 10, 3, 8
    X = rs.uniform(-3, 3, size=(N, d))
    y = X[:, 0] * X[:, 1] + np.sin(X[:, 2]) + 0.1 * rs.standard_normal(N)
    pool = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2)), un('cos', leaf(4)), un('square', leaf(5)), bi('+', leaf(6), leaf(7)),
            un('ln', un('exp', leaf(3)), 0.7, -0.2), un('cubic', leaf(8)), leaf(9)]
    # (a context for every chain at once: four chain groups over the same long slices, below)
    plain = DeviceContext(X, y, K=K, n_chains=C, max_batch=256)
    pi = plain.info()
    assert (pi["row_pass"], pi["tape_groups"], pi["row_slices"], pi["blocks_per_slice"]) == ("k_tile1a", 4, 98, 7), pi
    plain.close()
    ctx = DeviceContext(X, y, K=K, n_chains=C, max_batch=256, typical_chains=2, typical_batch=64)
    try:
        info = ctx.info()
        assert (info["row_pass"], info["tape_groups"], info["row_slices"], info["blocks_per_slice"]) == ("k_tile1a", 1, 98, 7), info
        cur = []
        for c in range(C):
            trees = [pool[(c + k) % len(pool)] for k in range(K)]
            cur.append(trees)
            for k in range(K):
                ctx.set_current(c, k, flatten(trees[k]))
            ctx.refresh(c)
        cands = [bi('+', un('sin', leaf(i % d)), leaf((i + 3) % d)) if i % 3 else bi('*', leaf(i % d), un('cos', leaf((i + 1) % d)))
                 for i in range(256)]
        tapes = [flatten(t) for t in cands]
        chains = (np.arange(256) // 32).astype(np.int32)
        ks = rs.randint(K, size=256).astype(np.int32)
        sig = rs.uniform(0.5, 1.5, size=256)
        wide = ctx.score_batch(tapes, chains, ks, sig)
        for g in range(4):
            sel = np.arange(g * 64, (g + 1) * 64)
            narrow = ctx.score_batch([tapes[i] for i in sel], chains[sel], ks[sel], sig[sel])
            assert narrow.tobytes() == wide[sel].tobytes(), g
        df = pd.DataFrame(X)
        for i in range(0, 256, 37):
            c = int(chains[i])
            cols = [O.allcal(O.tree_from_json(spec_from_node(t)), df)[:, 0] for t in cur[c]]
            out = np.stack(cols, axis=1).copy()
            out[:, ks[i]] = O.allcal(O.tree_from_json(spec_from_node(cands[i])), df)[:, 0]
            want = O.yloglike(pd.Series(y), out, float(sig[i]))
            if wide["rank"][i] == K and np.isfinite(want):
                assert abs(wide["loglik"][i] - want) <= 1e-6 * max(1.0, abs(want)), (i, wide["loglik"][i], want)
    finally:
        ctx.close()


def test_chain_groups_score_every_batch_shape_to_the_same_bytes():
    """Eight chains, up to 256 proposals per batch: the tape groups are CHAIN groups (batch chain i in group i mod 4), each
    staging its own chains' basis columns next to the features and y -- 17 columns instead of 35, so the eight-block slice
    fits LDS whole.  A proposal's bytes must not depend on the batch it travels in: all eight chains at once, two chains,
    one chain, five chains with gaps (groups of unequal size, one of them with a single chain) -- and the oracle's value."""
    import pandas as pd
    import bsr_oracle as O
    from conftest import spec_from_node
    rs = np.random.RandomState(6)
    N, d, K, C = 60_000, 8, 3, 8
    X = rs.uniform(-3, 3, size=(N, d))
    y = X[:, 0] * X[:, 1] + np.sin(X[:, 2]) + 0.1 * rs.standard_normal(N)
    pool = [bi('*', leaf(0), leaf(1)), un('sin', leaf(2)), un('cos', leaf(4)), un('square', leaf(5)), bi('+', leaf(6), leaf(7)),
            un('ln', un('exp', leaf(3)), 0.7, -0.2), un('cubic', leaf(1)), leaf(7)]
    ctx = DeviceContext(X, y, K=K, n_chains=C, max_batch=256)
    try:
        info = ctx.info()
        assert info["tape_groups"] == 4 and info["slices_whole"] and info["row_pass"] == "k_tile1a", info
        cur = []
        for c in range(C):
            trees = [pool[(2 * c + k) % len(pool)] for k in range(K)]
            cur.append(trees)
            for k in range(K):
                ctx.set_current(c, k, flatten(trees[k]))
            ctx.refresh(c)
        cands = [bi('+', un('sin', leaf(i % d)), leaf((i + 3) % d)) if i % 3 else bi('*', leaf(i % d), un('cos', leaf((i + 1) % d)))
                 for i in range(256)]
        tapes = [flatten(t) for t in cands]
        chains = (np.arange(256) // 32).astype(np.int32)
        ks = rs.randint(K, size=256).astype(np.int32)
        sig = rs.uniform(0.5, 1.5, size=256)
        wide = ctx.score_batch(tapes, chains, ks, sig)
        assert (wide["rank"] == K).sum() > 200
        shapes = [np.arange(0, 64), np.arange(96, 128), np.concatenate([np.arange(32, 64), np.arange(128, 160), np.arange(224, 256)]),
                  np.concatenate([np.arange(0, 5), np.arange(70, 100), np.arange(130, 131), np.arange(200, 256)]),
                  np.arange(255, -1, -1)]
        for sel in shapes:
            got = ctx.score_batch([tapes[i] for i in sel], chains[sel], ks[sel], sig[sel])
            assert got.tobytes() == wide[sel].tobytes(), (len(sel), sel[:3])
        df = pd.DataFrame(X)
        n_checked = 0
        for i in range(0, 256, 23):
            c = int(chains[i])
            cols = [O.allcal(O.tree_from_json(spec_from_node(t)), df)[:, 0] for t in cur[c]]
            out = np.stack(cols, axis=1).copy()
            out[:, ks[i]] = O.allcal(O.tree_from_json(spec_from_node(cands[i])), df)[:, 0]
            want = O.yloglike(pd.Series(y), out, float(sig[i]))
            if wide["rank"][i] == K and np.isfinite(want):
                assert abs(wide["loglik"][i] - want) <= 1e-6 * max(1.0, abs(want)), (i, wide["loglik"][i], want)
                n_checked += 1
        assert n_checked >= 6
    finally:
        ctx.close()
