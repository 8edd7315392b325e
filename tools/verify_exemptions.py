#!/usr/bin/env python3
"""The pinned exemptions of the GPU tests' 1e-6 log-likelihood gate, verified in the BUILD container (CPU, oracle only).

tests/golden/exemption_allow.json names, per fixture, the proposals (trace fixtures g5_trace_*: proposal indices) whose
device log-likelihood misses the fixture's by more than 1e-6 relative -- taken from a GPU run in discover mode
(BSR_EXEMPT_DISCOVER=1; gpurun_out/exemptions.json).  A proposal may be on that list only if its value is chaotic at the
ulp level: the ORACLE's own log-likelihood must move by more than 1e-7 relative when X is perturbed by one ulp (no two
libm builds agree on such a tree).  This script measures that spread for every listed proposal, here, once, and writes
it next to the index; it fails if a listed proposal is not chaotic.  The GPU tests then only check membership: nothing
about the gate depends on the numpy build of the GPU box.

    python tools/verify_exemptions.py [discover.json]     # (re)writes tests/golden/exemption_allow.json
"""
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsr_oracle as O

GOLDEN = os.path.join(ROOT, "tests", "golden")
ALLOW = os.path.join(GOLDEN, "exemption_allow.json")


def trace_spread(name, idx):
    g = json.load(open(os.path.join(GOLDEN, "g5_trace_%s.json" % name)))
    dat = np.load(os.path.join(GOLDEN, "g5_trace_%s.npz" % name))
    X, y = dat["X"], dat["y"]
    cur = [t for t in g["init_trees"]]
    for i, ref in enumerate(g["props"]):
        if i == idx:
            break
        if ref["accepted"]:
            cur[ref["count"]] = ref["proposed"]
    ref = g["props"][idx]

    def cols(Xp):
        Xd = pd.DataFrame(Xp)
        out = []
        for k, r in enumerate(cur):
            t = ref["proposed"] if k == ref["count"] else r
            with np.errstate(all="ignore"):
                out.append(O.allcal(O.tree_from_json(t), Xd)[:, 0])
        return np.stack(out, axis=1)
    sig = float(ref["new_sigma"])
    vals = []
    for eps in (0.0, 2.0 ** -52, -2.0 ** -52, 2.0 ** -51):
        with np.errstate(all="ignore"):
            vals.append(O.yloglike(np.asarray(y), cols(X * (1.0 + eps)), sig))
    spread = max(abs(v - vals[0]) for v in vals[1:]) / abs(vals[0])
    return float(spread), O.express(O.tree_from_json(ref["proposed"]))


def main():
    allow = json.load(open(ALLOW)) if os.path.exists(ALLOW) else {}
    if len(sys.argv) > 1:
        disc = json.load(open(sys.argv[1]))
        for k, v in disc.items():
            if v.get("ids"):
                allow[k] = {"ids": sorted(v["ids"]), "detail": v.get("detail")}
    ok = True
    for k in sorted(allow):
        if k.startswith("trace "):
            name = k[len("trace "):]
            ev = {}
            for i in allow[k]["ids"]:
                spread, expr = trace_spread(name, i)
                ev[str(i)] = {"oracle_relative_spread_under_one_ulp_of_X": spread, "proposed": expr}
                good = spread > 1e-7
                ok = ok and good
                print("%-28s proposal %4d  spread %.3e  %s  %s" % (k, i, spread, "chaotic" if good else "NOT CHAOTIC", expr[:80]))
            allow[k]["evidence"] = ev
        elif k.startswith("score_batch_vs_oracle"):
            fx = json.load(open(os.path.join(GOLDEN, "g9_score_batch.json")))["cases"][k[len("score_batch_vs_oracle "):]]
            for i in allow[k]["ids"]:
                good = bool(fx["chaotic"][i])
                ok = ok and good
                print("%-28s proposal %4d  fixture flag chaotic=%r  %s" % (k[:28], i, good, fx["express"][i][:80]))
            allow[k]["evidence"] = "tests/golden/g9_score_batch.json: chaotic[i], measured by tools/gen_golden_scores.py"
        else:
            print("%-28s ids %r (pinned by index; the chains' final state is checked to 1e-9 by the test itself)" % (k[:28], allow[k]["ids"]))
    allow["_about"] = ("per fixture: the proposals / chains the GPU tests may exempt from their tight bound; written from a "
                       "discover run on the GPU (BSR_EXEMPT_DISCOVER=1), verified by tools/verify_exemptions.py in the build container")
    with open(ALLOW, "w") as f:
        json.dump(allow, f, indent=1, sort_keys=True)
    if not ok:
        raise SystemExit("a listed proposal is not ulp-chaotic: it must not be exempt")


if __name__ == "__main__":
    main()
