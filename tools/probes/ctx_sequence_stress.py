"""Determinism stress: a sequence of freshly created contexts (K = 8, an ill-conditioned chain state whose scores
move with the last bit of anything that is off) must refresh and score byte-identically every time.

    python tools/probes/ctx_sequence_stress.py R [NB=2] [BSR_SELFDUP=0] ...   (GPU box; prints one line per context)
"""
import os, sys
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import numpy as np
import bsr_oracle as O
from conftest import node_from_spec, spec_from_node
from bsr.tape import flatten
from bsr.device import DeviceContext
from bsr.node import Node
N, d, K = 40000, 6, 8
B = 64
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
for j in range(min(K, 4)):
    tapes[j] = flatten(trees[j]); ks[j] = j
neg = Node(0)
neg.type, neg.operator, neg.left = 1, 'neg', node_from_spec(spec_from_node(trees[0]))
neg.left.parent = neg
tapes[5] = flatten(neg); ks[5] = 0
sig = rs.uniform(0.5, 2.0, size=B)
zeros = np.zeros(B, np.int32)
def run(env):
    os.environ.pop("BSR_SELFDUP", None)
    os.environ.update(env)
    c = DeviceContext(X, y, K=K, n_chains=1, max_batch=B)
    for k in range(K):
        c.set_current(0, k, flatten(trees[k]))
    c.refresh(0)
    out = c.score_batch(tapes, zeros, ks, sig).copy()
    c.close()
    return out
base = run({})
def trial(name, envs, reps):
    bad = 0
    for rep in range(reps):
        for env in envs:
            try:
                o = run(dict(env))
            except Exception as e:
                print(name, rep, env, "ERROR", str(e)[:80], flush=True)
                bad += 1
                continue
            if o.tobytes() != base.tobytes():
                d = np.nonzero(o["loglik"].view(np.uint64) != base["loglik"].view(np.uint64))[0]
                print(name, rep, env, "DIFFERENT at", len(d), "max rel %.3g" % np.abs(o["loglik"] / base["loglik"] - 1).max(), flush=True)
                bad += 1
    print(name, "bad", bad, "of", reps * len(envs), flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "R"
if which == "R":
    extra0 = dict(kv.split("=") for kv in sys.argv[2:])
    print("extra", extra0)
    for rep in range(12):
        extra = dict(extra0)
        os.environ.pop("BSR_SELFDUP", None)
        nb = int(extra.pop("NB", 2)) if "NB" in extra else 2
        os.environ.update(extra)
        c = DeviceContext(X, y, K=K, n_chains=1, max_batch=B)
        ev = c.eval_tapes([flatten(trees[k]) for k in range(K)])[0]
        cs = [float(np.nansum(np.abs(ev[k]))) for k in range(K)]
        for k in range(K):
            c.set_current(0, k, flatten(trees[k]))
        info = c.refresh(0)
        try:
            cur = c.get_current(0)
            curs = ["%.12g" % float(np.nansum(np.abs(cur[k]))) for k in range(K)]
            evs = ["%.12g" % v for v in cs]
            badk = [k for k in range(K) if curs[k] != evs[k]]
        except Exception as e:
            badk = str(e)
        outs = [c.score_batch(tapes, zeros, ks, sig).copy() for _ in range(nb)]
        c.close()
        print(rep, "refresh sse_old %.17g" % info["sse_old"], "score == base", [o.tobytes() == base.tobytes() for o in outs], "cur != eval at", badk, flush=True)
        for o in outs[:1]:
            if o.tobytes() != base.tobytes():
                dd = np.nonzero(o["loglik"].view(np.uint64) != base["loglik"].view(np.uint64))[0]
                print("    differ at", dd.tolist()[:20], "n", len(dd), "rel", np.abs(o["loglik"][dd] / base["loglik"][dd] - 1)[:6], "sse", o["sse"][dd][:3], base["sse"][dd][:3], "flags", o["flags"][dd][:6], base["flags"][dd][:6], "smin", o["smin"][dd][:3], base["smin"][dd][:3])
