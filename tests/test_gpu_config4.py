"""BASELINE configs[3] on the GPU box: config 4's per-GPU share (8 chains, N=100k, d=10, K=3, native sampler with
its worker threads) against single-chain runs and the CPU oracle; the sharded product entry; bench.py's own
multi-process launch.  The box has one GPU: several ranks share it with BSR_SHARE_DEVICE=1, which swaps only the
transport of the gather (RCCL refuses two ranks on one device) -- the RCCL calls themselves are covered with a
one-rank communicator here and with N ranks by the driver's scaling run."""
import json
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import spec_from_node

pytestmark = pytest.mark.gpu

import bsr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c4_data(N=100_000, d=10):
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    return X, y


def _summ(rec):
    from bsr.node import Express
    return ([Express(t) for t in rec["roots"]], rec["n_props"], rec["n_accept"], rec["n_rank_rejects"])


def test_config4_share_equals_single_chain_runs_and_the_oracle():
    """8 chains advanced together (worker threads on, 32 speculative proposals each per launch) give exactly the
    chains that run one at a time; chain 0 is the oracle's chain under the same seed (codes/bsr_class.py:99-273)."""
    from bsr import dist as D
    from bsr.sharded import run_rank
    X, y = _c4_data()
    K, val = 3, 40
    seeds = [1000 + c for c in range(8)]
    raw, stats = run_rank(X, y, K, seeds, rank=0, world=1, device=0, batch=32, val=val, chains_per_launch=8)
    together = [D.unpack_record(r) for r in raw]
    assert [r["chain"] for r in together] == list(range(8))
    assert stats["proposals"] == sum(r["n_props"] for r in together) and stats["chains"] == 8
    for c in range(8):
        raw1, _ = run_rank(X, y, K, [seeds[c]], rank=0, world=1, device=0, batch=32, val=val, chains_per_launch=1)
        one = D.unpack_record(raw1[0])
        assert _summ(one) == _summ(together[c]), c
        assert np.array_equal(one["beta"], together[c]["beta"]), c          # same arithmetic whatever shares the launch
        assert one["errs"] == together[c]["errs"], c
    # the oracle's chain 0 (vectorised flavour: same values as the faithful one, see tests/test_oracle_golden.py)
    np.random.seed(seeds[0])
    with np.errstate(all="ignore"):
        ref = O.run_chain(pd.DataFrame(X), pd.Series(y), K=K, val=val, faithful=False)
    got = together[0]
    assert [O.express(t) for t in ref["roots"]] == _summ(got)[0]
    assert ref["n_props"] == got["n_props"]
    assert len(ref["errs"]) == len(got["errs"]) == got["n_accept"]
    assert np.allclose(ref["errs"], got["errs"], rtol=1e-6)
    assert np.allclose(ref["beta"].reshape(-1), got["beta"].reshape(-1), rtol=1e-5, atol=1e-8)


def test_sharded_fit_product_entry_matches_in_process_fit():
    """BSR(devices=[0]) -- one child process per listed GPU, native sampler, gather -- returns what the in-process
    fit with the same per-chain seeds returns (roots_, betas_, train_err_ of codes/bsr_class.py:270-276)."""
    from bsr import BSR
    from bsr.node import Express
    X, y = _c4_data(N=5000, d=4)
    seeds = [2000 + c for c in range(5)]
    here = BSR(treeNum=3, itrNum=5, val=40, chain_seeds=seeds, chains_per_launch=4, batch=16)
    here.fit(X, y)
    far = BSR(treeNum=3, itrNum=5, val=40, chain_seeds=seeds, chains_per_launch=4, batch=16, devices=[0])
    far.fit(X, y)
    assert len(far.roots_) == 5
    for c in range(5):
        assert [Express(t) for t in far.roots_[c]] == [Express(t) for t in here.roots_[c]], c
        assert np.array_equal(far.betas_[c], here.betas_[c]), c
        assert far.train_err_[c] == here.train_err_[c], c
    assert far.stats_["proposals"] == here.stats_["proposals"] and far.model() == here.model()
    assert np.array_equal(far.predict(X[:50]), here.predict(X[:50]))


def test_two_ranks_sharing_the_device_give_the_one_rank_result():
    """World 2 (chains c % 2, both ranks on this box's only GPU, gather through the rendezvous directory) == world 1."""
    from bsr.node import Express
    from bsr.sharded import fit_sharded
    X, y = _c4_data(N=4000, d=3)
    seeds = [3000 + c for c in range(5)]
    one = fit_sharded(X, y, K=3, seeds=seeds, devices=[0], batch=16, val=30, chains_per_launch=3)
    two = fit_sharded(X, y, K=3, seeds=seeds, devices=[0, 0], batch=16, val=30, chains_per_launch=3,
                      env_extra={"BSR_SHARE_DEVICE": "1"})
    assert [r["chain"] for r in two] == [0, 1, 2, 3, 4]
    for a, b in zip(one, two):
        assert [Express(t) for t in a["roots"]] == [Express(t) for t in b["roots"]]
        assert np.array_equal(a["beta"], b["beta"]) and a["errs"] == b["errs"] and a["n_props"] == b["n_props"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two child ranks, one JSON line with n_gpus 2 and the gathered
    record count; the parent never touches the GPU."""
    env = dict(os.environ, BSR_SHARE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
                        "--cpu-sample", "0", "--extras", "0", "--min-time", "0", "--burnin", "60"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gathered_records"] == 2 and out["steps"] == 8
    assert out["config"]["proposals_per_step_per_gpu"] == 64 and out["value"] > 0
    assert out["roofline"]["frac"] > 0 and out["roofline"]["kernel_us"] > 0


def test_eight_ranks_on_one_device_complete_with_eight_records():
    """De-risking the first 8-GPU run: `BSR_SHARE_DEVICE=1 python bench.py --gpus 8 --extras 0` on the box's one GPU --
    eight child ranks under whatever CPU quota the box grants, each with the threads the library's CPU budget allows
    it, all of them through launch, rendezvous, barrier, timing reduction and gather; one JSON line with eight
    gathered records and a step time per rank.  (Only the gather's transport differs from the real thing: RCCL refuses
    eight ranks on one device.)"""
    env = dict(os.environ, BSR_SHARE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "8", "--warmup", "2",
                        "--cpu-sample", "0", "--extras", "0", "--min-time", "0", "--burnin", "40"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["gathered_records"] == 8
    pr = out["per_rank"]
    assert sorted(p["rank"] for p in pr) == list(range(8)) and all(p["ms_per_step"] > 0 for p in pr)
    # the ranks share the quota: nobody starts more submission threads than its share allows (two from six CPUs up)
    assert all(p["submit_threads"] <= (2 if p["cpu_budget"] >= 6 else 1 if p["cpu_budget"] >= 3 else 0) or
               p["submit_threads"] <= 2 for p in pr)
    assert out["verified"]["byte_identical"]


@pytest.mark.parametrize("n_chains,groups", [(4, 4), (8, 4), (8, 2), (16, 4), (8, 1), (4, 0)])
def test_the_score_memo_changes_nothing_a_chain_does(monkeypatch, n_chains, groups):
    """The native sampler answers exact repeats of a candidate in an unchanged chain state from a table (rank and SSE kept,
    the log-likelihood formed on the host for the proposal's own sigma: codes/funcs.py:1162-1173).  Chains with the memo
    and without it: the same accepted trees, proposal / accept / rank-rejection counts, Beta and RMSE history -- and a good
    share of the proposals is answered from the table (tools/memo_probe.py measured 40-49 % repeats in long chains; these
    short ones -- sixty accepted samples -- repeat less: 7 % here).
    Groups that hold SEVERAL chains (8 chains in 4 or 2 groups, 16 in 4: what `sharded.run_rank` and the bench run) are
    the case round 5's advisor broke: an accept answered from the memo used to resubmit a one-tape batch on the lane's
    slot while later chains of the same batch still committed by their index in it.  groups = 0: the traced,
    single-threaded ticket path (slot = -1, bsr_commit through the last waited batch)."""
    from bsr.chain import DeviceScorer
    from bsr.native import NativeEngine
    X, y = _c4_data()
    K, val = 3, 60
    out = {}
    if groups > 0:
        monkeypatch.setenv("BSR_ENGINE_GROUPS", str(groups))
    for memo in ("1", "0"):
        monkeypatch.setenv("BSR_ENGINE_MEMO", memo)
        scorer = DeviceScorer(X, y, K, n_chains=n_chains, max_batch=32 * n_chains)
        eng = NativeEngine(scorer.ctx, n_chains, X.shape[1], val=val)
        for c in range(n_chains):
            eng.seed(c, 2000 + c)
            eng.init_chain(c)
        if groups == 0:
            eng.run(batch_per_chain=32, trace_cap=200000)
        else:
            eng.run(batch_per_chain=32)
        res = [eng.result(c) for c in range(n_chains)]
        stats = [eng.memo_stats(c) for c in range(n_chains)]
        eng.close()
        scorer.close()
        from bsr.node import Express
        out[memo] = ([([Express(t) for t in r["roots"]], r["n_props"], r["n_accept"], r["n_rank_rejects"],
                       r["beta"].tobytes(), tuple(r["errs"])) for r in res], stats)
    assert out["1"][0] == out["0"][0]
    hits = sum(s[0] for s in out["1"][1])
    looks = sum(s[1] for s in out["1"][1])
    assert sum(s[0] for s in out["0"][1]) == 0
    assert looks > 0 and hits / looks > 0.03, (hits, looks)


@pytest.mark.parametrize("n_chains,groups,helpers", [(8, 4, "4"), (8, 2, "6"), (16, 4, "4"), (6, 1, "5")])
def test_helper_threads_change_nothing_a_chain_does(monkeypatch, n_chains, groups, helpers):
    """A group of several chains deals the per-chain halves of its cycle (generate, consume) to helper threads of its own
    (csrc/bsr_engine.hip: Group::ctl, round 6); a chain's candidates depend on its own trees and random stream only, so the
    chains must end exactly where they end without helpers -- trees, counters, Beta, RMSE history (codes/bsr_class.py:99:
    the reference runs its chains one after the other)."""
    from bsr.chain import DeviceScorer
    from bsr.native import NativeEngine
    from bsr.node import Express
    X, y = _c4_data()
    K, val = 3, 60
    out = {}
    monkeypatch.setenv("BSR_ENGINE_GROUPS", str(groups))
    for h in (helpers, "0"):
        monkeypatch.setenv("BSR_ENGINE_HELPERS", h)
        scorer = DeviceScorer(X, y, K, n_chains=n_chains, max_batch=32 * n_chains)
        eng = NativeEngine(scorer.ctx, n_chains, X.shape[1], val=val)
        for c in range(n_chains):
            eng.seed(c, 3000 + c)
            eng.init_chain(c)
        eng.run(batch_per_chain=32)
        res = [eng.result(c) for c in range(n_chains)]
        eng.close()
        scorer.close()
        out[h] = [([Express(t) for t in r["roots"]], r["n_props"], r["n_accept"], r["n_rank_rejects"],
                   r["beta"].tobytes(), tuple(r["errs"])) for r in res]
    assert out[helpers] == out["0"]
    assert sum(r[2] for r in out["0"]) > 0 and all(r[1] > 0 for r in out["0"])
