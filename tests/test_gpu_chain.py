"""End-to-end parity on the GPU: the reference's golden chain traces and BSR.fit run through the HIP scorer."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, farr, load_golden, node_from_spec, unf

pytestmark = pytest.mark.gpu

import bsr_oracle as O
from test_host_driver import TRACES, replay_trace


def _device_scorer(X, y, K):
    from bsr.chain import DeviceScorer
    return DeviceScorer(X, y, K, n_chains=1, max_batch=64)


@pytest.mark.parametrize("name", TRACES)
@pytest.mark.parametrize("batch", [1, 32])
def test_reference_traces_through_the_hip_scorer(name, batch):
    """Accepted-tree sequence, actions, rank-gate decisions and RNG position bit-exact; log-likelihoods to 1e-6."""
    if batch == 1 and name in ("f1_s0", "synth_K8_s1001"):
        pytest.skip("long trace covered at batch 32")
    ch = replay_trace(name, _device_scorer, batch, 1e-6)
    ch.scorer.close()


@pytest.mark.parametrize("name", TRACES)
@pytest.mark.parametrize("batch", [1, 32])
def test_reference_traces_through_the_native_engine(name, batch):
    """The C++ sampler (bsr_engine_*) replays the reference's traces: actions, decisions, RNG position bit-exact."""
    from bsr import _lib
    from bsr.device import DeviceContext
    from bsr.native import NativeEngine
    from test_host_driver import tree_hash_from_spec
    if batch == 1 and name in ("f1_s0", "synth_K8_s1001"):
        pytest.skip("long trace covered at batch 32")
    g = load_golden("g5_trace_%s.json" % name)
    dat = np.load(os.path.join(GOLDEN, "g5_trace_%s.npz" % name))
    X, y = dat["X"], dat["y"]
    K = g["K"]
    ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=64)
    eng = NativeEngine(ctx, 1, X.shape[1], val=g["val"], y_is_series=not name.endswith("yarr"))
    if "weights" in g:
        eng.set_ops(g["ops"], g["weights"])
    eng.seed(0, g["seed"])
    eng.init_chain(0)
    tr = eng.run(batch_per_chain=batch, max_props=g["n_props"] if g["truncated"] else -1, trace_cap=g["n_props"] + 8)
    assert len(tr) == g["n_props"]
    n_chaotic = 0
    chaotic_ids = []
    from conftest import exempt_allowed
    from test_host_driver import _ulp_sensitive
    allowed = exempt_allowed("trace %s" % name)                        # pinned per fixture (None: discover mode)
    cur_nodes = [node_from_spec(t) for t in g["init_trees"]]         # the chain's current trees, replayed from the golden
    for i, (ref, got) in enumerate(zip(g["props"], tr)):
        tag = "%s batch %d proposal %d" % (name, batch, i)
        assert ref["count"] == got["count"], tag
        assert ref["action"] == _lib.ACTIONS[got["action"]], tag
        assert ref["change"] == _lib.CHANGES[got["change"]], tag
        for key in ("Q", "Qinv", "new_sa2", "new_sb2"):
            assert abs(unf(ref[key]) - got[key]) <= 1e-12 * abs(unf(ref[key])), (tag, key)
        assert ref["rank"] == got["rank"] or (ref["rank"] < K and got["rank"] < K), tag
        h, n = tree_hash_from_spec(ref["proposed"])
        assert (h, n) == (int(got["tree_hash"]), int(got["n_nodes"])), tag
        if ref["rank"] == K:
            assert abs(unf(ref["new_sigma"]) - got["new_sigma"]) <= 1e-13 * got["new_sigma"], tag
            for key in ("yllstar", "yll"):
                want = unf(ref[key])
                if np.isfinite(want) and not abs(want - got[key]) <= 1e-6 * abs(want):
                    # allowed only for a proposal whose value is chaotic at the ulp level: the oracle's own number must
                    # move under a one-ulp perturbation of X
                    # move under a one-ulp perturbation of X -- established in the build container for the pinned
                    # proposals (tests/golden/exemption_allow.json), measured live only in discover mode
                    assert key == "yllstar", (tag, key, want, got[key])
                    if allowed is None:
                        rec = {"cur_roots": cur_nodes, "proposed": node_from_spec(ref["proposed"]), "count": ref["count"],
                               "new_sigma": unf(ref["new_sigma"])}
                        assert _ulp_sensitive(rec, X, y, 1e-6), (tag, key, want, got[key])
                    else:
                        assert i in allowed, (tag, key, want, got[key], sorted(allowed))
                    n_chaotic += 1
                    chaotic_ids.append(i)
        assert ref["accepted"] == bool(got["accepted"]), tag
        if ref["accepted"]:
            cur_nodes[ref["count"]] = node_from_spec(ref["proposed"])
    from conftest import note_exempt
    note_exempt("trace %s" % name, n_chaotic, len(tr), ids=chaotic_ids)
    st = eng.get_numpy_state(0)
    import zlib
    last = g["props"][-1]["rng"]
    assert int(st[2]) == last["pos"] and int(zlib.crc32(st[1].tobytes())) == last["crc"], name
    if not g["truncated"]:
        from bsr.node import Express
        r = eng.result(0)
        assert [Express(t) for t in r["roots"]] == g["final_models"]
        assert np.allclose(r["beta"].reshape(-1), farr(g["betas"]), rtol=1e-6, atol=1e-9)
        assert np.allclose(r["errs"], farr(g["train_err"]), rtol=1e-8)
    eng.close()
    ctx.close()


# what a loose chain's RMSE history may miss the reference's by: the two chains that take this route (pinned by index in
# tests/golden/exemption_allow.json) measured 4.0e-3 and 3.5e-4 (profiles/r05_exemptions_discover.json) -- sin / cos of
# 1e12-sized arguments, where the libm build decides the last digits; their final state is pinned to 1e-9 below
LOOSE_RMSE_DEV = 1e-2


@pytest.mark.parametrize("engine", ["native", "python"])
def test_bsr_fit_f1_matches_reference_end_to_end(engine):
    """BSR(3,50).fit on f1, np.random.seed(0): every chain's model strings, proposal counts, Beta, RMSE history,
    predict() -- the reference's config 1 (20 061 proposals)."""
    from bsr import BSR
    g = load_golden("g6_fit_f1.json")
    X = pd.DataFrame(np.array(g["X"], dtype=np.float64))
    y = pd.Series(farr(g["y"]))
    np.random.seed(0)
    np.random.uniform(0.1, 5.9, 100)
    np.random.uniform(0.1, 5.9, 100)
    est = BSR(treeNum=3, itrNum=50, engine=engine)
    assert est.fit(X, y) is None
    assert est.stats_["proposals"] == g["total_props"]
    from bsr.node import Express
    loose = 0
    loose_ids, loose_dev = [], {}
    for c in range(50):
        assert [Express(t) for t in est.roots_[c]] == g["models"][c], c          # accepted trees: exact
        assert len(est.train_err_[c]) == len(g["train_err"][c]), c
        want_b, want_e = farr(g["betas"][c]), farr(g["train_err"][c])
        got_b, got_e = np.asarray(est.betas_[c]).reshape(-1), np.asarray(est.train_err_[c])
        dev = 0.0
        if len(want_e):
            dev = float(np.max(np.abs(got_e - want_e) / np.abs(want_e)))
        dev_b = float(np.max(np.abs(got_b - want_b) / (np.abs(want_b) + 1e-6 * np.max(np.abs(want_b)))))
        # values agree to 1e-7 except for chains that visit trees which are chaotic at the ulp level
        # (sin/cos of 1e12-sized arguments): there only the libm build decides the last digits
        if dev > 1e-7 or dev_b > 1e-5:
            loose += 1
            loose_ids.append(c)
            loose_dev[c] = (dev, dev_b)
            assert dev < LOOSE_RMSE_DEV and np.isfinite(dev_b), (c, dev, dev_b, g["models"][c])
            # ... and ONLY libm's last digits: the chain's final fit against the oracle's intercept fit of the DEVICE's own
            # columns of the same trees (the solver in isolation, as tests/test_gpu_kernels.py does for the scores) to
            # 1e-9 -- a refresh or solver regression cannot hide in the allowance above.  (A chain that ended through the
            # plateau break keeps its pre-accept trees, codes/bsr_class.py:180-182 vs :204: no final state to compare.)
            if True:   # (every loose chain, whatever its length)
                import bsr as _bsr
                import bsr_oracle as _O
                with np.errstate(all="ignore"):
                    cols = np.hstack([np.asarray(_bsr.allcal(t, X)).reshape(-1, 1) for t in est.roots_[c]])
                    ob, orm = _O.intercept_fit(np.asarray(y, dtype=np.float64), cols)
                ob = np.asarray(ob).reshape(-1)
                assert np.allclose(got_b, ob, rtol=1e-9, atol=1e-9 * np.max(np.abs(ob))), (c, got_b, ob)
                if len(got_e):
                    assert abs(got_e[-1] - orm) <= 1e-9 * max(abs(orm), 1e-300), (c, got_e[-1], orm)
    from conftest import note_exempt
    note_exempt("config 1 fit: chains whose Beta/RMSE history miss 1e-7 / 1e-5", loose, 50, ids=loose_ids,
                detail={str(c): [float(v) for v in loose_dev[c]] for c in loose_ids})
    assert est.model() == g["model_last"]
    assert est.complexity() == g["complexity"]
    grid = np.array(g["grid"])
    assert np.allclose(est.predict(grid)[:, 0], farr(g["predict_grid"]), rtol=1e-6, atol=1e-8)
    assert np.allclose(est.predict(grid, last_ind=2)[:, 0], farr(g["predict_grid_last2"]), rtol=1e-6, atol=1e-8)
    m = np.random.get_state()
    import zlib
    assert int(m[2]) == g["rng_end"]["pos"] and int(zlib.crc32(m[1].tobytes())) == g["rng_end"]["crc"]


def test_module_functions_allcal_yloglike_newprop():
    """bsr.allcal / bsr.ylogLike / bsr.newProp keep the reference's call signatures and results."""
    import bsr
    g = load_golden("g5_trace_f1_s7.json")
    dat = np.load(os.path.join(GOLDEN, "g5_trace_f1_s7.npz"))
    X = pd.DataFrame(dat["X"])
    y = pd.Series(dat["y"])
    Ops, W, T = list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY)
    roots = [node_from_spec(s) for s in g["init_trees"]]
    cols = []
    for s, r in zip(g["init_trees"], roots):
        out = bsr.allcal(r, X)
        assert out.shape == (len(y), 1)
        with np.errstate(all="ignore"):
            want = O.allcal(O.tree_from_json(s), X)
        assert np.allclose(out, want, rtol=1e-12, atol=1e-12)
        cols.append(out[:, 0])
    cols = np.stack(cols, axis=1)
    ll = bsr.ylogLike(y, cols, 0.9)
    assert abs(ll - O.yloglike(y, cols, 0.9)) <= 1e-9 * abs(ll)
    # replay the first proposals of the trace through newProp, one call per proposal like BSR.fit does
    np.random.seed(g["seed"])
    from bsr import rng
    from bsr import proposal as P
    sigma = rng.invgamma_rvs(1)
    cur, siga, sigb = [], [], []
    for k in range(3):
        root = bsr.Node(0)
        sa, sb = rng.invgamma_rvs(1), rng.invgamma_rvs(1)
        bsr.grow(root, 2, Ops, W, T, -1, sa, sb)
        cur.append(root)
        siga.append(sa)
        sigb.append(sb)
    n_acc = 0
    for i, ref in enumerate(g["props"][:120]):
        k = ref["count"]
        res, sigma, Root, siga[k], sigb[k] = bsr.newProp(cur, k, sigma, y, X, 2, Ops, W, T, -1, siga[k], sigb[k])
        assert res == ref["accepted"], i
        assert abs(sigma - unf(ref["sigma_out"])) <= 1e-13 * sigma, i
        if res:
            n_acc += 1
            cur[k] = Root
    assert n_acc == sum(1 for r in g["props"][:120] if r["accepted"])


@pytest.mark.parametrize("engine", ["native", "python"])
def test_parallel_chains_equal_single_chain_runs(engine):
    """Chains seeded individually and advanced several per launch give exactly the single-chain results."""
    from bsr import BSR
    from bsr.node import Express
    rs = np.random.RandomState(3)
    X = rs.uniform(-3, 3, size=(400, 3))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(400)
    seeds = [1000 + c for c in range(6)]
    par = BSR(treeNum=3, itrNum=6, val=60, chain_seeds=seeds, chains_per_launch=4, batch=16, engine=engine)
    par.fit(X, y)
    for c, s in enumerate(seeds):
        np.random.seed(s)
        one = BSR(treeNum=3, itrNum=1, val=60, engine="python")
        one.fit(X, y)
        assert [Express(t) for t in one.roots_[0]] == [Express(t) for t in par.roots_[c]], c
        if engine == "python":      # same code path: bit for bit
            assert np.array_equal(one.betas_[0], par.betas_[c]), c
            assert one.train_err_[0] == par.train_err_[c], c
        else:                       # the C++ sampler's invgamma quantile differs from scipy's in the last ulps
            assert np.allclose(one.betas_[0], par.betas_[c], rtol=1e-5, atol=1e-9), c
            assert np.allclose(one.train_err_[0], par.train_err_[c], rtol=1e-7), c


def test_native_engine_worker_threads_do_not_change_results(monkeypatch):
    """The native sampler with one worker thread per chain group gives bit-identical chains to the one-thread run."""
    from bsr.device import DeviceContext
    from bsr.native import NativeEngine
    from bsr.node import Express
    rs = np.random.RandomState(21)
    N, d, K, C = 20000, 5, 3, 8
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    runs = []
    for threads in ("1", "0"):
        monkeypatch.setenv("BSR_ENGINE_THREADS", threads)
        ctx = DeviceContext(X, y, K=K, n_chains=C, max_batch=32 * C)
        eng = NativeEngine(ctx, C, d, val=10 ** 9)
        eng.set_nan_policy(True)
        for c in range(C):
            eng.seed(c, 500 + c)
            eng.init_chain(c)
        eng.run(batch_per_chain=32, max_props=1500)
        res = [eng.result(c, current=True) for c in range(C)]
        runs.append([([Express(t) for t in r["roots"]], r["beta"].tobytes(), np.asarray(r["errs"]).tobytes(),
                      r["n_props"], r["n_accept"], r["n_rank_rejects"], eng.get_numpy_state(c)[1].tobytes())
                     for c, r in enumerate(res)])
        eng.close()
        ctx.close()
    assert runs[0] == runs[1]
    assert sum(r[4] for r in runs[0]) > 0          # some proposals were accepted on the way


def test_predict_all_and_model_file_on_device(tmp_path):
    """predict_all() row -i == predict(last_ind=i); a saved and reloaded model predicts bit-identically."""
    from bsr import BSR
    rs = np.random.RandomState(8)
    X = rs.uniform(-3, 3, size=(300, 3))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(300)
    est = BSR(treeNum=3, itrNum=5, val=40, chain_seeds=[11, 12, 13, 14, 15], chains_per_launch=5, batch=16)
    est.fit(X, y)
    Xt = rs.uniform(-3, 3, size=(77, 3))
    allp = est.predict_all(Xt)
    assert allp.shape == (5, 77)
    for i in range(1, 6):
        assert np.array_equal(allp[-i], est.predict(Xt, last_ind=i)[:, 0]), i
    path = str(tmp_path / "m.json")
    est.save(path)
    back = BSR.load(path)
    assert back.model() == est.model()
    assert np.array_equal(back.predict_all(Xt), allp)
    with np.errstate(all="ignore"):
        import pandas as pd
        cols = np.stack([O.allcal(O.tree_from_json(__import__("conftest").spec_from_node(t)), pd.DataFrame(Xt))[:, 0]
                         for t in est.roots_[-1]], axis=1)
    want = np.concatenate((np.ones((77, 1)), cols), axis=1) @ est.betas_[-1]
    assert np.allclose(allp[-1], want[:, 0], rtol=1e-9, atol=1e-9)


def test_rccl_gather_through_the_c_abi_single_rank():
    """bsr_comm_unique_id / init / allgather with a one-rank communicator: the RCCL path of the accepted-tree gather
    (the multi-rank exchange itself is covered with gloo in tests/test_dist_gloo.py and by the driver's N>1 bench runs)."""
    from bsr import dist
    from bsr.device import DeviceContext
    from bsr.node import Express
    rs = np.random.RandomState(2)
    X = rs.uniform(-3, 3, size=(500, 3))
    y = X[:, 0] * X[:, 1] + 0.1 * rs.standard_normal(500)
    ctx = DeviceContext(X, y, K=2, n_chains=1, max_batch=4)
    uid = DeviceContext.comm_unique_id()
    g = dist.RcclGather(ctx, 1, 0, uid)
    g2 = load_golden("g2_grow.json")
    trees = [node_from_spec(c["tree"]) for c in g2["cases"][:2]]
    rec = dist.pack_record(5, trees, np.array([[0.5], [1.5], [-2.0]]), 0.9, [1.0, 0.75], 10, 2)
    got = dist.gather_chains(g, [rec], 2)
    assert len(got) == 1 and got[0]["chain"] == 5
    assert [Express(t) for t in got[0]["roots"]] == [Express(t) for t in trees]
    assert np.array_equal(got[0]["beta"].reshape(-1), [0.5, 1.5, -2.0])
    assert (got[0]["n_props"], got[0]["n_accept"], got[0]["n_errs"]) == (10, 2, 2)
    ctx.close()


def test_device_side_mh_step_against_the_host(monkeypatch):
    """log-ratio assembly, accept test and first-event scan on the device (k_events).
    (1) through the C ABI with hand-made terms: the event of every span equals a numpy restatement of
        codes/funcs.py:1226-1306 on the scores of the same batch, logR bit for bit;
    (2) the native sampler with the device step on gives the chains it gives with the step off, and with
        BSR_ENGINE_VERIFY_MH=1 the host recomputes every decision and logR and finds them identical."""
    from bsr import _lib
    from bsr.device import DeviceContext
    from bsr.native import NativeEngine
    from bsr.node import Express
    from bsr.tape import flatten, pack
    rs = np.random.RandomState(5)
    N, d, K = 3000, 4, 3
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    ctx = DeviceContext(X, y, K=K, n_chains=2, max_batch=64)
    np.random.seed(3)
    trees = []
    while len(trees) < 2 * K + 48:
        root = O.ONode(0)
        O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
        if O.count_nodes(root) < 40:
            trees.append(node_from_spec(__import__("conftest").spec_from_node(root)))
    for c in range(2):
        for k in range(K):
            ctx.set_current(c, k, flatten(trees[c * K + k]))
        ctx.refresh(c)
    tapes = [flatten(t) for t in trees[2 * K:]]
    tapes[5] = flatten(trees[1])                       # repeats a sibling of chain 0: the rank gate rejects it
    B = len(tapes)
    chains = np.array([0] * 20 + [1] * 28, np.int32)
    span_off = np.array([0, 20, 48], np.int32)
    ks = (np.arange(B) % K).astype(np.int32)
    ks[5] = 0
    sig = rs.uniform(0.5, 2.0, size=B)
    rows, off = pack(tapes)
    plain = ctx.score_batch(tapes, chains, ks, sig).copy()
    for trial in range(6):
        terms = np.zeros((B, 8))
        terms[:, 0] = np.where(np.isfinite(plain["loglik"]), plain["loglik"], 0.0) + rs.uniform(-3, 6, size=B)  # yll
        terms[:, 1:7] = rs.uniform(-2, 2, size=(B, 6))
        terms[:, 7] = np.log(rs.uniform(size=B))
        flags = rs.randint(0, 2, size=B).astype(np.int32)            # JUMP or not
        if trial % 2:
            flags[rs.randint(0, B, size=6)] |= _lib.MH_NO_UNIFORM
        if trial == 5:
            terms[:, 0] += 50.0                                     # nothing is accepted: the gate decides
        t = ctx.score_submit_mh(rows, off, chains, ks, sig, terms, flags, span_off)
        out = np.zeros(B, dtype=plain.dtype)
        ev = ctx.score_wait_mh(t, out)
        assert out.tobytes() == plain.tobytes()
        for sp in range(2):
            want = (span_off[sp + 1] - span_off[sp], _lib.EV_NONE, np.nan)
            for i in range(span_off[sp], span_off[sp + 1]):
                no_u = bool(flags[i] & _lib.MH_NO_UNIFORM)
                if out["rank"][i] < K:
                    if no_u:
                        continue
                    want = (i - span_off[sp], _lib.EV_GATE, np.nan)
                    break
                tt = terms[i]
                log_y = out["loglik"][i] - tt[0]
                logR = (log_y + tt[1] + tt[2] + tt[3] + tt[4]) if flags[i] & _lib.MH_JUMP else (log_y + tt[1] + tt[2])
                logR = logR + tt[5] - tt[6]
                if no_u:
                    want = (i - span_off[sp], _lib.EV_GATE_PASSED, logR)
                    break
                if not (tt[7] >= min(logR, 0)):
                    want = (i - span_off[sp], _lib.EV_ACCEPT, logR)
                    break
            got = (int(ev["index"][sp]), int(ev["kind"][sp]), float(ev["logR"][sp]))
            assert got[:2] == want[:2], (trial, sp, got, want)
            assert got[2] == want[2] or (np.isnan(got[2]) and np.isnan(want[2])), (trial, sp, got, want)
    ctx.close()
    # (2) the sampler
    runs = {}
    for mode, env in (("host", {"BSR_ENGINE_DEVICE_MH": "0"}), ("device", {"BSR_ENGINE_DEVICE_MH": "1"}),
                      ("verify", {"BSR_ENGINE_DEVICE_MH": "1", "BSR_ENGINE_VERIFY_MH": "1"})):
        for k_, v_ in (("BSR_ENGINE_DEVICE_MH", None), ("BSR_ENGINE_VERIFY_MH", None)):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        ctx = DeviceContext(X, y, K=K, n_chains=6, max_batch=6 * 32)
        eng = NativeEngine(ctx, 6, d, val=10 ** 9)
        eng.set_nan_policy(True)
        for c in range(6):
            eng.seed(c, 700 + c)
            eng.init_chain(c)
        eng.run(batch_per_chain=32, max_props=3000)
        res = [eng.result(c, current=True) for c in range(6)]
        runs[mode] = [([Express(t) for t in r["roots"]], r["beta"].tobytes(), r["n_props"], r["n_accept"],
                       r["n_rank_rejects"], eng.get_numpy_state(c)[1].tobytes()) for c, r in enumerate(res)]
        eng.close()
        ctx.close()
    assert runs["host"] == runs["device"] == runs["verify"]
    assert sum(r[3] for r in runs["host"]) > 0
