import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd")); sys.path.insert(0, ROOT)
import numpy as np
from bench import synth
from bsr import _lib
from bsr.chain import Chain, DeviceScorer, run_chains
from bsr.tape import pack
N, d, K, B = 100000, 10, 3, 64
X, y = synth(N, d, seed=0)
scorer = DeviceScorer(X, y, K, n_chains=1, max_batch=B); ctx = scorer.ctx
np.random.seed(1000); ch = Chain(0, scorer, N, d, K, val=10**9)
run_chains([ch], scorer, batch_per_chain=B, max_props=300)
packed = []
for _ in range(220):
    tapes, chs, ks, sig = [], [], [], []
    for cd in ch.generate(B):
        tapes.append(cd.tape); chs.append(0); ks.append(cd.k); sig.append(cd.new_sigma)
    ch.rng_state = ch._end_state
    rows, off = pack(tapes)
    packed.append((rows, off, np.array(chs, np.int32), np.array(ks, np.int32), np.array(sig), np.zeros(B, dtype=_lib.SCORE_DTYPE)))
for depth in (1, 2, 3, 4):
    for r in packed[:20]: ctx.score_packed(*r)
    ts = tw = 0.0; tickets = []
    t0 = time.perf_counter()
    for r in packed[20:]:
        a = time.perf_counter(); tickets.append((ctx.score_submit(r[0], r[1], r[2], r[3], r[4]), r)); ts += time.perf_counter() - a
        if len(tickets) >= depth:
            t, rr = tickets.pop(0); a = time.perf_counter(); ctx.score_wait(t, rr[5]); tw += time.perf_counter() - a
    while tickets:
        t, rr = tickets.pop(0); a = time.perf_counter(); ctx.score_wait(t, rr[5]); tw += time.perf_counter() - a
    el = time.perf_counter() - t0
    n = len(packed) - 20
    print("depth %d: %.1f us/step  (submit %.1f us, wait %.1f us, other python %.1f us)" % (depth, el / n * 1e6, ts / n * 1e6, tw / n * 1e6, (el - ts - tw) / n * 1e6))
