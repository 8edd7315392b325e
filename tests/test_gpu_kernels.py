"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.  Needs an MI355X."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, farr, load_golden, node_from_spec, spec_from_node, unf

pytestmark = pytest.mark.gpu

import bsr_oracle as O


def _dev():
    from bsr.device import DeviceContext
    return DeviceContext


def _close(a, b, rtol, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    same_nan = np.isnan(a) == np.isnan(b)
    same_inf = (np.isinf(a) == np.isinf(b)) & (~np.isinf(a) | (np.sign(a) == np.sign(b)))
    fin = np.isfinite(a) & np.isfinite(b)
    ok = np.abs(a[fin] - b[fin]) <= atol + rtol * np.abs(b[fin])
    return bool(same_nan.all() and same_inf.all() and ok.all())


EXACT_OPS = {"terminal", "inv", "ln", "neg", "square", "+", "*"}


def test_g1_edge_semantics_on_device():
    from bsr.tape import flatten
    g = load_golden("g1_edge.json")
    X = np.stack([farr(g["x0"]), farr(g["x1"])], axis=1)
    ctx = _dev()(X, None, max_batch=32)
    tapes = [flatten(node_from_spec(c["tree"])) for c in g["cases"]]
    cols, maxabs, flags = ctx.eval_tapes(tapes)
    for c, col, fl in zip(g["cases"], cols, flags):
        want = farr(c["out"])
        if c["name"] in EXACT_OPS:
            assert _close(col, want, 0.0), c["name"]            # bit-exact
            assert np.array_equal(np.signbit(col[~np.isnan(col)]), np.signbit(want[~np.isnan(want)])), c["name"]
        else:
            assert _close(col, want, 4e-16, 1e-320), (c["name"], col, want)   # <= 2 ulp for sin/cos/exp/cubic
        assert bool(fl & 1) == bool(np.isinf(want).any()), c["name"]
        assert bool(fl & 2) == bool(np.isnan(want).any()), c["name"]
    ctx.close()


def test_g2_grown_trees_on_device():
    from bsr.tape import flatten, unflatten
    g = load_golden("g2_grow.json")
    for c in g["cases"]:
        np.random.seed(c["seed"])
        X = np.random.uniform(-3, 3, size=(64, c["d"]))
        ctx = _dev()(X, None, max_batch=4)
        root = node_from_spec(c["tree"])
        tape = flatten(root)
        assert len(tape) == c["n_nodes"]
        cols, maxabs, flags = ctx.eval_tapes([tape, flatten(unflatten(tape))])
        want = farr(c["out"])
        fin = np.isfinite(want)
        tol = 1e-13 * max(1.0, float(np.max(np.abs(want[fin]))) if fin.any() else 1.0)
        assert _close(cols[0], want, 1e-12, tol), (c["seed"], c["express"])
        assert _close(cols[1], cols[0], 0.0), c["seed"]
        if fin.all():
            assert abs(maxabs[0] - np.max(np.abs(cols[0]))) == 0.0
        ctx.close()


def test_g3_yloglike_on_device():
    from bsr.device import yloglike_device
    g = load_golden("g3_yloglike.json")
    for c in g["cases"]:
        y = farr(c["y"])
        Om = np.array([[unf(v) for v in row] for row in c["O"]], dtype=np.float64)
        r = yloglike_device(y, Om, c["sigma"], skipna=c["y_is_series"])
        assert r["rank"] == c["rank"], (c["name"], r)
        want = unf(c["loglik"])
        if c["rank"] == c["K"]:
            tol = 1e-6 if "collinear" in c["name"] else 1e-9
            assert abs(r["loglik"] - want) <= tol * abs(want), (c["name"], r["loglik"], want)
            assert abs(r["scale"] - unf(c["scale"])) <= 1e-15 * unf(c["scale"]), c["name"]
            assert abs(r["sse"] - unf(c["sse"])) <= tol * unf(c["sse"]) + 1e-12 * float(y @ y), c["name"]


def _rand_trees(rs_seed, n, d, min_nodes=1):
    out = []
    np.random.seed(rs_seed)
    while len(out) < n:
        root = O.ONode(0)
        O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
        if O.count_nodes(root) >= min_nodes and O.count_nodes(root) < 200:
            out.append(root)
    return out


@pytest.mark.parametrize("N,d,K,B,seed", [(3000, 6, 3, 48, 1), (1000, 4, 8, 40, 2), (777, 3, 1, 24, 3),
                                          (5000, 10, 2, 64, 4), (130, 2, 3, 64, 5), (800, 4, 4, 32, 6),
                                          (800, 5, 5, 32, 7), (600, 4, 6, 32, 8), (600, 5, 7, 32, 9)])
def test_score_batch_vs_oracle(N, d, K, B, seed):
    """Random current trees + random candidate trees: rank gate, SSE, log-likelihood and Beta against the oracle."""
    from bsr.tape import flatten
    rs = np.random.RandomState(seed)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1 % d] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1 % d] - 1)) + 0.1 * rs.standard_normal(N)
    ctx = _dev()(X, y, K=K, n_chains=2, max_batch=B)
    trees = _rand_trees(100 + seed, 2 * K + B, d)
    cur = [trees[:K], trees[K:2 * K]]
    cands = trees[2 * K:]
    Xdf = pd.DataFrame(X)
    cur_cols = []
    for c in range(2):
        for k in range(K):
            ctx.set_current(c, k, flatten(node_from_spec(spec_from_node(cur[c][k]))))
        with np.errstate(all="ignore"):
            cur_cols.append(np.stack([O.allcal(t, Xdf)[:, 0] for t in cur[c]], axis=1))
        info = ctx.refresh(c)
        if np.all(np.isfinite(cur_cols[c])):
            _, sse_o, scale_o, _ = O.yloglike_parts(y, cur_cols[c], 1.0)
            assert abs(info["sse_old"] - sse_o) <= 1e-9 * sse_o, (c, info["sse_old"], sse_o)
            assert abs(info["scale_old"] - scale_o) <= 1e-12 * scale_o
        got = ctx.get_current(c)
        assert _close(got.T, cur_cols[c], 1e-12, 1e-13 * (1 + np.nanmax(np.abs(np.where(np.isfinite(cur_cols[c]), cur_cols[c], 0)))))
    chains = rs.randint(0, 2, size=B)
    ks = rs.randint(0, K, size=B)
    sig = rs.uniform(0.3, 3.0, size=B)
    tapes = [flatten(node_from_spec(spec_from_node(t))) for t in cands]
    res = ctx.score_batch(tapes, chains, ks, sig)
    dev_cols, _, _ = ctx.eval_tapes(tapes)          # candidate columns as the device evaluates them
    res2 = ctx.score_batch(tapes, chains, ks, sig)  # eval_tapes reuses the candidate slots: rescore
    assert res.tobytes() == res2.tobytes()           # deterministic, bit for bit
    dev_cur = [ctx.get_current(c).T for c in range(2)]
    n_full = n_chaotic = 0
    chaotic_ids = []
    # The oracle's scores of exactly this batch come from a fixture written in the build container
    # (tools/gen_golden_scores.py -> tests/golden/g9_score_batch.json): the 1e-6 end-to-end gate, and which proposals are
    # ulp-chaotic enough to be exempt from it (the oracle's own value moves under a one-ulp perturbation of X, measured
    # THERE), do not depend on the numpy build of the box the GPU runs in.
    fx = load_golden("g9_score_batch.json")["cases"]["N=%d d=%d K=%d B=%d seed=%d" % (N, d, K, B, seed)]
    for i in range(B):
        want = {"rank": fx["rank"][i], "loglik": unf(fx["loglik"][i]), "scale": unf(fx["scale"][i])}
        assert fx["express"][i] == O.express(cands[i])
        tag = "proposal %d %s rank %r" % (i, O.express(cands[i]), want["rank"])
        assert int(res["rank"][i]) == want["rank"] or (want["rank"] < K and 0 <= res["rank"][i] < K), (tag, res[i])
        assert (res["rank"][i] == K) == (want["rank"] == K), (tag, res[i])
        if want["rank"] == K:
            n_full += 1
            # (1) north-star bound against the oracle end to end: 1e-6 relative on the log-likelihood.  Exception: trees
            #     that are chaotic at the ulp level (cos(exp(x^6)): the value depends on the libm build), flagged in the fixture
            if not abs(res["loglik"][i] - want["loglik"]) <= 1e-6 * abs(want["loglik"]):
                assert fx["chaotic"][i], (tag, res[i], want)
                n_chaotic += 1
                chaotic_ids.append(i)
            assert abs(res["scale"][i] - want["scale"]) <= 1e-12 * want["scale"], tag
            # (2) solver in isolation: oracle fed with the device's own columns (removes libm ulp differences
            #     that ill-conditioned trees such as sin(exp(1/x)) amplify)
            w2 = O.score_proposal(dev_cur[chains[i]], ks[i], dev_cols[i], y, sig[i])
            cond = res["smax"][i] / res["smin"][i]
            tol = 1e-10 if cond < 1e5 else 1e-7
            assert w2["rank"] == K, tag
            assert abs(res["loglik"][i] - w2["loglik"]) <= tol * abs(w2["loglik"]), (tag, cond, res[i], w2)
            assert abs(res["sse"][i] - w2["sse"]) <= tol * abs(w2["sse"]) + 1e-12 * float(y @ y), (tag, cond, res[i], w2)
            assert np.all(np.abs(res["beta"][i][:K] - w2["beta"]) <= 1e-6 * np.max(np.abs(w2["beta"])) * max(1.0, cond * 1e-6)), (tag, res[i], w2)
    assert n_full > 0
    from conftest import note_exempt
    note_exempt("score_batch_vs_oracle N=%d d=%d K=%d B=%d seed=%d" % (N, d, K, B, seed), n_chaotic, n_full, ids=chaotic_ids)
    # accept the first full-rank proposal: commit == set_current of the same tape
    i = int(np.argmax(res["rank"] == K))
    ctx.commit(int(chains[i]), int(ks[i]), i)
    info_a = ctx.refresh(int(chains[i]))
    beta_a, rmse_a = ctx.fit_beta(int(chains[i]))
    cols_a = ctx.get_current(int(chains[i]))
    ctx.set_current(int(chains[i]), int(ks[i]), tapes[i])
    info_b = ctx.refresh(int(chains[i]))
    beta_b, rmse_b = ctx.fit_beta(int(chains[i]))
    assert np.array_equal(cols_a, ctx.get_current(int(chains[i])))
    assert info_a["sse_old"] == info_b["sse_old"] and rmse_a == rmse_b and np.array_equal(beta_a, beta_b)
    with np.errstate(all="ignore"):
        bo, ro = O.intercept_fit(y, cols_a.T)
    assert abs(rmse_a - ro) <= 1e-9 * ro
    assert np.all(np.abs(beta_a - bo) <= 1e-6 * np.max(np.abs(bo)) + 1e-9 * np.abs(bo))
    ctx.close()


def test_extended_operator_table_on_device():
    """sub / div / log (semantics defined by the oracle) through tape, kernels and both samplers.
    Columns of random trees over the 13-operator table: bit-identical to the oracle where every opcode's device
    arithmetic is exactly numpy's (terminal, inv, ln, neg, square, +, *, sub, div), within 4 ulp otherwise; scores
    against the oracle; BSR(ops=..., op_weights=...) gives the same chains through the C++ and the Python sampler."""
    from bsr import BSR
    from bsr.node import Express
    from bsr.tape import flatten
    ops = list(O.OPS) + list(O.EXT_OPS)
    arity = list(O.OP_ARITY) + list(O.EXT_ARITY)
    w = [1.0 / len(ops)] * len(ops)
    N, d, K = 2000, 5, 3
    rs = np.random.RandomState(31)
    X = rs.uniform(-3, 3, size=(N, d))
    X[::97, 1] = 0.0                                     # exact zeros: the protected branches of div and log
    y = X[:, 0] - X[:, 1] / (1.5 + X[:, 2] ** 2) + np.log(np.abs(X[:, 3]) + 0.1) + 0.1 * rs.standard_normal(N)
    np.random.seed(77)
    trees = []
    while len(trees) < 120:
        root = O.ONode(0)
        O.grow(root, d, ops, w, arity, -1, 1.0, 1.0)
        names = {n.operator for n in O.preorder(root) if n.type > 0}
        if 2 <= O.count_nodes(root) < 60 and names & set(O.EXT_OPS):
            trees.append(root)
    ctx = _dev()(X, y, K=K, n_chains=1, max_batch=128)
    tapes = [flatten(node_from_spec(spec_from_node(t))) for t in trees]
    cols, maxabs, flags = ctx.eval_tapes(tapes)
    Xdf = pd.DataFrame(X)
    exact_ops = {'inv', 'ln', 'neg', 'square', '+', '*', 'sub', 'div'}
    n_exact = 0
    for i, t in enumerate(trees):
        with np.errstate(all="ignore"):
            want = O.allcal(t, Xdf)[:, 0]
        names = {n.operator for n in O.preorder(t) if n.type > 0}
        if names <= exact_ops:
            n_exact += 1
            assert np.array_equal(cols[i], want, equal_nan=True), (i, O.express(t))
        elif np.all(np.isfinite(want)) and not names & {'sin', 'cos', 'exp'}:      # log / cubic: a few ulp per node;
            err = np.abs(cols[i] - want) / np.maximum(np.abs(want), 1e-300)        # rows where a log sits near its zero
            assert np.quantile(err, 0.9) <= 1e-13 and np.median(err) <= 4e-16, (i, O.express(t), np.quantile(err, 0.9))
    assert n_exact >= 20
    # scoring: current trees and candidates from the same pool
    cur = trees[:K]
    for k in range(K):
        ctx.set_current(0, k, tapes[k])
    ctx.refresh(0)
    with np.errstate(all="ignore"):
        cur_cols = np.stack([O.allcal(t, Xdf)[:, 0] for t in cur], axis=1)
    B = len(trees) - K
    ks = (np.arange(B) % K).astype(np.int32)
    sig = rs.uniform(0.5, 2.0, size=B)
    res = ctx.score_batch(tapes[K:], np.zeros(B, np.int32), ks, sig)
    n_full = n_exempt = 0
    if np.all(np.isfinite(cur_cols)):
        for i in range(B):
            with np.errstate(all="ignore"):
                col = O.allcal(trees[K + i], Xdf)[:, 0]
            want = O.score_proposal(cur_cols, ks[i], col, y, sig[i])
            if want["rank"] == K:
                n_full += 1
                assert int(res["rank"][i]) == K, (i, O.express(trees[K + i]))
                if not abs(res["loglik"][i] - want["loglik"]) <= 1e-6 * abs(want["loglik"]):
                    # exempt only if the ORACLE's own value moves by more than the tolerance under a one-ulp change of X
                    with np.errstate(all="ignore"):
                        col2 = O.allcal(trees[K + i], pd.DataFrame(np.nextafter(X, np.inf)))[:, 0]
                    w2 = O.score_proposal(cur_cols, ks[i], col2, y, sig[i])
                    assert w2["rank"] != K or not abs(w2["loglik"] - want["loglik"]) <= 1e-6 * abs(want["loglik"]), \
                        (i, O.express(trees[K + i]), res["loglik"][i], want["loglik"])
                    n_exempt += 1
            else:
                assert int(res["rank"][i]) == want["rank"] or (want["rank"] >= 0 and 0 <= res["rank"][i] < K), i
    ctx.close()
    from conftest import note_exempt
    note_exempt("extended operator table, score vs oracle", n_exempt, n_full)
    # both samplers with a non-default table (weights favour the extensions)
    ww = [1.0] * 10 + [3.0, 3.0, 3.0]
    runs = []
    for engine in ("native", "python"):
        est = BSR(treeNum=2, itrNum=3, val=40, chain_seeds=[5, 6, 7], chains_per_launch=3, batch=16, engine=engine,
                  ops=ops, op_weights=ww)
        est.fit(X, y)
        runs.append(([[Express(t) for t in r] for r in est.roots_], est.stats_["proposals"]))
    assert runs[0] == runs[1]
    assert any(("log(" in m or ")-(" in m or ")/[" in m) for r in runs[0][0] for m in r)
