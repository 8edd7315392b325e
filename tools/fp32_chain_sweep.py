#!/usr/bin/env python3
"""BASELINE configs[4]'s fp32-vs-fp64 tolerance sweep at the level of the CHAIN (run on the GPU box):

    python tools/fp32_chain_sweep.py [--Ns 10000,100000,1000000] [--d 50] [--consumed 2000] [--out profiles/...json]

One chain (seed 1000, the reference's sampler restated in bsr/chain.py) runs on an fp64 context; an fp32 context (f32
storage and tree arithmetic, f64 accumulation) shadows it -- same current trees, same proposals.  For every proposal
the chain CONSUMES, the fp32 side's verdict is worked out as the reference would (codes/funcs.py:1226-1228 rank gate,
:1298-1304 accept test: only the two log-likelihoods of logR depend on the dtype) and compared:

  * first_divergence: index of the first consumed proposal whose decision (gate or accept) differs -- from there on an
    fp32 chain is on another trajectory (another accepted-tree sequence) than the fp64 / reference one;
  * rank_flips / accept_flips per 1000 consumed proposals along the fp64 trajectory (the state is re-synchronised
    after every flip, so the rate is per proposal, not cumulative);
  * max and median |dloglik| (nats) and relative, over full-rank proposals.

The accept margin |logR - log u| is O(1) nats for almost every proposal: fp32 is usable as long as |dloglik| stays far
below that."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, ROOT)

import numpy as np

from bench import synth
from bsr import proposal as P
from bsr.chain import Chain, DeviceScorer, Scorer, run_chains


class Shadowed(Scorer):
    """fp64 scorer that drives the chain, fp32 scorer kept in the same state."""

    def __init__(self, X, y, K, batch):
        self.a = DeviceScorer(X, y, K, n_chains=1, max_batch=batch, dtype="f64")
        self.b = DeviceScorer(X, y, K, n_chains=1, max_batch=batch, dtype="f32")
        self.K, self.max_batch = K, batch
        self.ctx = self.a.ctx
        self.last_b = None
        self.pos = 0
        self.info_a = self.info_b = None

    def set_tree(self, chain, k, tape):
        self.a.set_tree(chain, k, tape)
        self.b.set_tree(chain, k, tape)

    def refresh(self, chain):
        self.info_a = self.a.refresh(chain)
        self.info_b = self.b.refresh(chain)
        return self.info_a

    def score(self, tapes, chains, ks, sigmas):
        ra = self.a.score(tapes, chains, ks, sigmas)
        self.last_b = self.b.score(tapes, chains, ks, sigmas).copy()
        self.pos = 0
        return ra

    def commit(self, chain, k, slot):
        self.a.commit(chain, k, slot)
        self.b.commit(chain, k, slot)

    def fit_beta(self, chain):
        self.b.fit_beta(chain)
        return self.a.fit_beta(chain)

    def close(self):
        self.a.close()
        self.b.close()


def sweep_one(N, d, K, batch, consumed):
    X, y = synth(N, d, seed=0)
    sc = Shadowed(X, y, K, batch)
    st = {"n": 0, "rank_flips": 0, "accept_flips": 0, "first": None, "dll": [], "rel": [], "accepts": 0, "gate": 0}
    ch_box = []

    def on_consumed(rec):
        ch = ch_box[0]
        i = st["n"]
        st["n"] += 1
        rb = sc.last_b[sc.pos]
        sc.pos += 1
        ra_rank = int(rec["rank"])
        flip = False
        if ra_rank < K:
            st["gate"] += 1
        if (ra_rank < K) != (int(rb["rank"]) < K):
            st["rank_flips"] += 1
            st["to_deficient"] = st.get("to_deficient", 0) + (1 if ra_rank == K else 0)   # fp64 full rank, fp32 not
            flip = True
        elif ra_rank == K:
            ll_a, ll_b = float(rec["yllstar"]), float(rb["loglik"])
            sig = st.get("sigma_now", ch.sigma)
            # the old state's log-likelihood on either side (codes/funcs.py:1233-1235), same sigma
            dyll = -(sc.info_b["sse_old"] - sc.info_a["sse_old"]) / (2 * sig * sig)
            logR_b = rec["logR"] + (ll_b - ll_a) - dyll
            acc_b = bool(P.accept_test(logR_b, rec["u"]))
            st["dll"].append(abs(ll_b - ll_a))
            st["rel"].append(abs(ll_b - ll_a) / abs(ll_a))
            st["accepts"] += int(bool(rec["accepted"]))
            if acc_b != bool(rec["accepted"]):
                st["accept_flips"] += 1
                flip = True
        if flip and st["first"] is None:
            st["first"] = i
        st["sigma_now"] = ch.sigma      # (an accept changes sigma AFTER its own test: keep the value the next test sees)

    np.random.seed(1000)
    ch = Chain(0, sc, N, d, K, val=10 ** 9, trace=on_consumed)
    ch_box.append(ch)
    st["sigma_now"] = ch.sigma
    run_chains([ch], sc, batch_per_chain=batch, max_props=consumed)
    sc.close()
    dll = np.array(st["dll"]) if st["dll"] else np.zeros(1)
    rel = np.array(st["rel"]) if st["rel"] else np.zeros(1)
    n = max(1, st["n"])
    return {"N": N, "d": d, "K": K, "consumed": st["n"], "gate_rejects": st["gate"], "accepts_f64": st["accepts"],
            "first_divergence": st["first"], "rank_flips_per_1000": 1000.0 * st["rank_flips"] / n,
            "accept_flips_per_1000": 1000.0 * st["accept_flips"] / n, "rank_flips": st["rank_flips"],
            "accept_flips": st["accept_flips"], "rank_flips_fp32_deficient_fp64_full": st.get("to_deficient", 0),
            "abs_dloglik_max": float(dll.max()),
            "abs_dloglik_median": float(np.median(dll)), "rel_dloglik_max": float(rel.max()),
            "rel_dloglik_median": float(np.median(rel))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--Ns", default="10000,100000,1000000")
    ap.add_argument("--d", type=int, default=50)
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--consumed", type=int, default=2000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = []
    for N in [int(v) for v in a.Ns.split(",")]:
        r = sweep_one(N, a.d, a.K, a.batch, a.consumed)
        rows.append(r)
        print(json.dumps(r), flush=True)
    out = {"what": "fp32 context shadowing an fp64 chain (seed 1000), decisions per consumed proposal", "rows": rows}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
