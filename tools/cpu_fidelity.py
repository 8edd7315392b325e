#!/usr/bin/env python3
"""CPU-baseline fidelity: the oracle's reference-faithful flavour timed next to
the REAL reference on the same seeded proposals, in the build container (needs /root/reference; never runs on the
GPU box).

For each seed the chain of codes/bsr_class.py:116-142 is set up and `n` proposals are pushed through the reference's
`newProp` (codes/funcs.py:1184) and, from the same RNG state and chain state, through `oracle.newprop(faithful=True)`.
Both consume numpy's global stream identically, so they see the same proposals; the outcomes are compared and the
wall-clock ratio is printed.  The figure the GPU bench reports as `cpu_baseline` (scoring work only: K+1 tree
evaluations, rank gate, two ylogLike passes -- no proposal generation, no deep copies) is timed on the same
proposals as well.

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_fidelity.py [--n 240] [--N 100000] [--d 10] [--K 3]
"""
import argparse
import copy
import json
import os
import sys
import time
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = "/tmp/refshim"
os.makedirs(SHIM, exist_ok=True)
if not os.path.islink(os.path.join(SHIM, "bsr")):
    os.symlink("/root/reference/codes", os.path.join(SHIM, "bsr"))
sys.path.insert(0, SHIM)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
warnings.filterwarnings("ignore")

import numpy as np
import pandas as pd
from scipy.stats import invgamma

import bsr.funcs as RF
import bsr_oracle as O

OPS = ['inv', 'ln', 'neg', 'sin', 'cos', 'exp', 'square', 'cubic', '+', '*']
OPW = [1.0 / len(OPS)] * len(OPS)
OPT = [1, 1, 1, 1, 1, 1, 1, 1, 2, 2]


def synth(N, d, seed=0):
    rs = np.random.RandomState(seed)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    return X, y


def start_chain(mod_node, mod_grow, K, d, beta=-1):
    """codes/bsr_class.py:123-142 with either implementation's Node/grow."""
    sigma = invgamma.rvs(1)
    roots, sa, sb = [], [], []
    for _ in range(K):
        root = mod_node(0)
        a = invgamma.rvs(1)
        b = invgamma.rvs(1)
        mod_grow(root, d, OPS, OPW, OPT, beta, a, b)
        roots.append(root)
        sa.append(a)
        sb.append(b)
    return sigma, roots, sa, sb


def run(newprop, node_cls, grow_fn, express, X, y, K, n, seed, **kw):
    np.random.seed(seed)
    d = X.shape[1]
    sigma, roots, sa, sb = start_chain(node_cls, grow_fn, K, d)
    t0 = time.perf_counter()
    out = []
    done = 0
    with np.errstate(all="ignore"):
        while done < n:
            for k in range(K):
                try:
                    res, sigma, root, a, b = newprop(roots, k, sigma, y, X, d, OPS, OPW, OPT, -1, sa[k], sb[k], **kw)
                except np.linalg.LinAlgError:
                    out.append("linalg")
                    return out, time.perf_counter() - t0, done
                sa[k], sb[k] = a, b
                if res:
                    roots[k] = copy.deepcopy(root)
                out.append((bool(res), express(root), float(sigma)))
                done += 1
                if done >= n:
                    break
    return out, time.perf_counter() - t0, done


def scoring_only(X, y, K, n, seed):
    """What bench.py's cpu_baseline leg does per proposal, on the same proposals (oracle objects)."""
    np.random.seed(seed)
    d = X.shape[1]
    sigma, roots, sa, sb = start_chain(O.ONode, O.grow, K, d)
    props = []
    with np.errstate(all="ignore"):
        while len(props) < n:
            for k in range(K):
                tr = {}
                try:
                    res, sigma, root, a, b = O.newprop(roots, k, sigma, y, X, d, OPS, OPW, OPT, -1, sa[k], sb[k],
                                                       faithful=False, trace=tr)
                except np.linalg.LinAlgError:
                    break
                props.append((k, copy.deepcopy(tr["proposed"]), [copy.deepcopy(r) for r in roots], tr["new_sigma"], sigma))
                sa[k], sb[k] = a, b
                if res:
                    roots[k] = copy.deepcopy(root)
                if len(props) >= n:
                    break
            else:
                continue
            break
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        for k, cand, cur, new_sigma, old_sigma in props:
            new_o = np.zeros((len(y), K))
            old_o = np.zeros((len(y), K))
            for j in range(K):
                if j == k:
                    new_o[:, j] = O.allcal(cand, X, True)[:, 0]
                    old_o[:, j] = O.allcal(cur[j], X, True)[:, 0]
                else:
                    col = O.allcal(cur[j], X, True)[:, 0]
                    new_o[:, j] = col
                    old_o[:, j] = col
            try:
                full = np.linalg.matrix_rank(new_o) == K
            except np.linalg.LinAlgError:
                full = False
            if full:
                O.yloglike(y, new_o, float(new_sigma))
                O.yloglike(y, old_o, float(old_sigma))
    return len(props), time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=240, help="proposals per seed")
    ap.add_argument("--seeds", type=int, nargs="*", default=[1000, 1001])
    ap.add_argument("--N", type=int, default=100_000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--K", type=int, default=3)
    a = ap.parse_args()
    Xa, ya = synth(a.N, a.d)
    X, y = pd.DataFrame(Xa), pd.Series(ya)
    tot = {"ref": [0, 0.0], "oracle": [0, 0.0], "scoring": [0, 0.0]}
    same = True
    for seed in a.seeds:
        r_out, r_t, r_n = run(RF.newProp, RF.Node, RF.grow, RF.Express, X, y, a.K, a.n, seed)
        o_out, o_t, o_n = run(O.newprop, O.ONode, O.grow, O.express, X, y, a.K, a.n, seed, faithful=True)
        same &= (r_out == o_out)
        s_n, s_t = scoring_only(X, y, a.K, a.n, seed)
        tot["ref"][0] += r_n; tot["ref"][1] += r_t
        tot["oracle"][0] += o_n; tot["oracle"][1] += o_t
        tot["scoring"][0] += s_n; tot["scoring"][1] += s_t
        print("seed %d: reference %d proposals %.1f s (%.2f/s) | oracle faithful %.1f s (%.2f/s) | same outcomes: %s | "
              "scoring-only leg %d proposals %.1f s (%.2f/s)" %
              (seed, r_n, r_t, r_n / r_t, o_t, o_n / o_t, r_out == o_out, s_n, s_t, s_n / s_t), flush=True)
    res = {"N": a.N, "d": a.d, "K": a.K, "proposals": tot["ref"][0],
           "reference_newProp_per_s": tot["ref"][0] / tot["ref"][1],
           "oracle_faithful_newprop_per_s": tot["oracle"][0] / tot["oracle"][1],
           "ratio_oracle_over_reference": (tot["oracle"][0] / tot["oracle"][1]) / (tot["ref"][0] / tot["ref"][1]),
           "bench_scoring_only_leg_per_s": tot["scoring"][0] / tot["scoring"][1],
           "identical_outcomes": bool(same),
           "versions": {"numpy": np.__version__, "pandas": pd.__version__, "python": sys.version.split()[0]},
           "cpus": os.cpu_count()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
