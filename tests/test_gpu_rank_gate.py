"""G4: the rank gate `np.linalg.matrix_rank(new_outputs) < K` (codes/funcs.py:1226) on the inputs where it is
closest to its threshold -- `new_outputs` matrices captured at the reference's own call during seeded chains
(sigma_min / sigma_max within 1e3 x of max(N, K) eps), plus constructed near-repeats (1e-9 ... 1e-15), scale disparities
(1e9 ... 1e16) and exact sums, each with the rank the reference's call returned (tools/gen_golden.py g4).
The device never sees the N x K matrix as a whole: it decides from the (K+1) x K factor of the candidate against the
chain's orthonormal basis (csrc/bsr_solve.h) -- by bounds where the matrix is far from the threshold, by one-sided Jacobi
in the band around it.  Both tiers, every position k of the replaced tree.  Needs an MI355X."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


def _cases():
    meta = load_golden("g4_rank.json")["cases"]
    with np.load(os.path.join(GOLDEN, "g4_rank.npz")) as z:
        mats = [z["M%d" % i] for i in range(len(meta))]
    return meta, mats


def _leaf(f):
    from bsr.tape import NODE_DTYPE
    t = np.zeros(1, dtype=NODE_DTYPE)
    t["opcode"], t["left"], t["right"], t["feature"] = 10, -1, -1, f
    return t


@pytest.mark.parametrize("exact", ["0", "1"])
def test_g4_rank_gate_near_its_threshold(monkeypatch, exact):
    from bsr.device import DeviceContext
    monkeypatch.setenv("BSR_SOLVE_EXACT", exact)
    meta, mats = _cases()
    assert len(meta) >= 100 and sum(1 for m in meta if m["origin"].startswith("chain")) >= 40
    eps = np.finfo(np.float64).eps
    wrong, undecidable, n_bounds, n_checked = [], 0, 0, 0
    for ci, (m, M) in enumerate(zip(meta, mats)):
        N, K = M.shape
        rs = np.random.RandomState(ci)
        extra = rs.standard_normal(N) * np.max(np.abs(M))     # the tree that is replaced: any column of the matrix's scale
        X = np.concatenate([M, extra[:, None]], axis=1)
        y = rs.standard_normal(N)
        ctx = DeviceContext(X, y, K=K, n_chains=K, max_batch=K)
        for c in range(K):        # chain c: trees = the matrix's columns but tree c, which the proposal puts back
            for j in range(K):
                ctx.set_current(c, j, _leaf(K if j == c else j))
            ctx.refresh(c)
        res = ctx.score_batch([_leaf(c) for c in range(K)], np.arange(K), np.arange(K), np.full(K, 1.0))
        ctx.close()
        # LAPACK's singular values of the N x K matrix carry an absolute error of a few eps sigma_max themselves: a
        # ratio within that of the tolerance is decided by rounding on either side
        tol = max(N, K) * eps
        slack = 8 * eps / tol
        for c in range(K):
            n_checked += 1
            full = int(res["rank"][c]) == K
            n_bounds += int(bool(res["flags"][c] & 16))
            if abs(m["ratio_over_tol"] - 1.0) <= slack:
                undecidable += 1
                continue
            if full != (m["rank"] == K):
                wrong.append((ci, c, m["origin"], m["ratio_over_tol"], m["rank"], int(res["rank"][c]), int(res["flags"][c])))
            if full:
                assert np.isfinite(res["loglik"][c]) and np.isfinite(res["sse"][c]), (ci, c, res[c])
    assert not wrong, wrong[:10]
    assert undecidable <= 0.05 * n_checked, (undecidable, n_checked)
    if exact == "1":
        assert n_bounds == 0                      # every verdict from singular values
    else:
        assert n_bounds > 0.2 * n_checked          # far from the threshold the bounds settle it (most constructed cases)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8])
def test_the_fast_tier_agrees_with_the_jacobi_tier(monkeypatch, K):
    """csrc/bsr_solve.h: solve_fast (Givens QR of the nearly triangular factor, the gate by bounds, the ridge fit by
    rotations of [tau T; 1e-3 I]) against the one-sided Jacobi SVD that scored every proposal until round 5
    (BSR_SOLVE_EXACT=1), on random trees of the real generator: the same gate verdicts; log-likelihood, SSE and Beta of
    the full-rank proposals to 1e-10 (1e-7 beyond a condition number of 1e6); the exact tier's singular values inside
    the fast tier's bounds."""
    from bsr.device import DeviceContext
    from bsr.tape import flatten
    from conftest import node_from_spec, spec_from_node
    from test_gpu_kernels import _rand_trees
    N, d, B = 3000, 6, 96
    rs = np.random.RandomState(50 + K)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    trees = _rand_trees(300 + K, 2 * K + B, d)
    tapes = [flatten(node_from_spec(spec_from_node(t))) for t in trees]
    chains = rs.randint(0, 2, size=B)
    ks = rs.randint(0, K, size=B)
    sig = rs.uniform(0.3, 3.0, size=B)
    # a few candidates that repeat or nearly repeat a sibling / the tree they replace
    for i in range(0, 12):
        tapes[2 * K + i] = tapes[int(chains[i]) * K + int(rs.randint(K))]
    out = {}
    for exact in ("0", "1"):
        monkeypatch.setenv("BSR_SOLVE_EXACT", exact)
        ctx = DeviceContext(X, y, K=K, n_chains=2, max_batch=B)
        for c in range(2):
            for k in range(K):
                ctx.set_current(c, k, tapes[c * K + k])
            ctx.refresh(c)
        out[exact] = ctx.score_batch(tapes[2 * K:], chains, ks, sig)
        ctx.close()
    f, e = out["0"], out["1"]
    assert not (e["flags"] & 16).any()
    n_full = n_fast = 0
    for i in range(B):
        tag = (i, f[i], e[i])
        assert (f["rank"][i] == K) == (e["rank"][i] == K), tag
        assert f["rank"][i] == e["rank"][i] or (0 <= f["rank"][i] < K and 0 <= e["rank"][i] < K) or e["rank"][i] <= 0, tag
        if not (f["flags"][i] & 16):
            assert f[i].tobytes() == e[i].tobytes(), tag        # the band: the Jacobi tier itself
            continue
        n_fast += 1
        if e["rank"][i] == K:
            n_full += 1
            cond = e["smax"][i] / e["smin"][i]
            tol = 1e-10 if cond < 1e6 else 1e-7
            assert abs(f["loglik"][i] - e["loglik"][i]) <= tol * abs(e["loglik"][i]), tag
            assert abs(f["sse"][i] - e["sse"][i]) <= tol * abs(e["sse"][i]) + 1e-13 * float(y @ y), tag
            assert np.all(np.abs(f["beta"][i][:K] - e["beta"][i][:K]) <= 1e-9 * max(1.0, cond * 1e-3) * np.max(np.abs(e["beta"][i][:K])) + 1e-300), tag
            assert f["smin"][i] <= e["smin"][i] * (1 + 1e-9) and f["smax"][i] >= e["smax"][i] * (1 - 1e-9), tag
            assert f["smin"][i] >= e["smin"][i] / (K + 1e-9) and f["smax"][i] <= e["smax"][i] * (np.sqrt(K) + 1e-9), tag
        else:
            assert np.isnan(f["loglik"][i]) and np.isnan(f["sse"][i]), tag
    assert n_full >= B // 3 and n_fast >= 0.8 * B, (n_full, n_fast)
