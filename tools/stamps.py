#!/usr/bin/env python3
"""Per-phase cycle budget of a row-pass wave from in-kernel clock stamps (debug build with -DBSR_STAMPS).

Run on the GPU box:  bash tools/stamps.sh        (builds nothing; expects build_variants/lib_stamps.so)
Stamps per wave: [wave start] then per proposal [proposal start, sweep done x sweeps, reductions done]."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, ROOT)
import numpy as np
from bench import synth
from bsr import _lib
from bsr.chain import Chain, DeviceScorer, run_chains
from bsr.tape import pack

N, d, K, B = int(os.environ.get("N", 100000)), int(os.environ.get("D", 10)), 3, int(os.environ.get("B", 64))
X, y = synth(N, d, seed=0)
scorer = DeviceScorer(X, y, K, n_chains=1, max_batch=B)
ctx = scorer.ctx
np.random.seed(1000)
ch = Chain(0, scorer, N, d, K, val=10 ** 9)
run_chains([ch], scorer, batch_per_chain=B, max_props=300)
out = np.zeros(B, dtype=_lib.SCORE_DTYPE)
from bsr.node import Node
from bsr.tape import flatten
def leaf(f):
    n = Node(1); n.type = 0; n.feature = np.array([f]); return n
def un(op, c, a=None, b=None):
    n = Node(0); n.type, n.operator, n.left, n.a, n.b = 1, op, c, a, b; c.parent = n; return n
def bi(op, l, r):
    n = Node(0); n.type, n.operator, n.left, n.right = 2, op, l, r; l.parent = r.parent = n; return n
def chain_of(op, n):
    t = leaf(1)
    for _ in range(n):
        t = un(op, t, 0.9, 0.1) if op == 'ln' else un(op, t)
    return t
CASES = {"x1": lambda: leaf(1), "x1*x2": lambda: bi('*', leaf(1), leaf(2)), "neg4": lambda: chain_of('neg', 4),
         "neg16": lambda: chain_of('neg', 16), "sin": lambda: chain_of('sin', 1), "sin4": lambda: chain_of('sin', 4),
         "exp": lambda: chain_of('exp', 1), "4term": lambda: bi('+', bi('*', leaf(1), leaf(2)), bi('*', leaf(3), leaf(4)))}
case = os.environ.get("CASE")
for it in range(4):
    tapes, chs, ks, sig = [], [], [], []
    for cd in ch.generate(B):
        tapes.append(flatten(CASES[case]()) if case else cd.tape); chs.append(0); ks.append(cd.k); sig.append(cd.new_sigma)
    ch.rng_state = ch._end_state
    rows, off = pack(tapes)
    ctx.score_packed(rows, off, np.array(chs, np.int32), np.array(ks, np.int32), np.array(sig), out)
L = _lib.lib()
NW, NS = 16384, 48
buf = np.zeros(NW * NS, dtype=np.uint64)
fn = ctypes.CDLL(_lib.LIB_PATH).bsr_debug_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = fn(buf.ctypes.data, buf.size)
assert rc == 0, rc
print('nonzero stamps', int((buf > 0).sum()))
base = buf[buf > 0].min()
st = np.where(buf > 0, buf - base + 1, 0).reshape(NW, NS).astype(np.int64)
sweeps = int(os.environ.get("SWEEPS", 4))
per_prop = 1 + sweeps + 1
ctx.set_profiling(1); ctx.score_packed(rows, off, np.array(chs, np.int32), np.array(ks, np.int32), np.array(sig), out); print('kernel us', ctx.last_timing()[0])
rows_ = []
t_first = st[:, 0][st[:, 0] > 0].min()
starts, ends, setup, sweep_t, red = [], [], [], [], []
for w in range(NW):
    s = st[w]
    if s[0] == 0: continue
    n = int((s[:46] > 0).sum())
    nprop = (n - 1) // per_prop
    if nprop == 0: continue
    starts.append(s[0] - t_first)
    ends.append(s[1 + nprop * per_prop - 1] - t_first)
    for p in range(nprop):
        b = 1 + p * per_prop
        prev = s[b - 1]
        setup.append(s[b] - prev)
        for k in range(sweeps):
            sweep_t.append(s[b + 1 + k] - s[b + k])
        red.append(s[b + per_prop - 1] - s[b + sweeps])
f = lambda a: "mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (np.mean(a), np.median(a), np.percentile(a, 90), np.max(a))
print("case", case); # calibration: span of the stamped waves of XCD 0 (workgroup id % 8 == 0 share one clock) against the event time
raw = buf.reshape(NW, NS)
wall = raw[:, 46:48].astype(np.int64)
ok = (wall[:, 0] > 0) & (wall[:, 1] > wall[:, 0])
life_wall = (wall[ok, 1] - wall[ok, 0])
span_wall = wall[ok, 1].max() - wall[ok, 0].min()
print("wall clock (10 ns ticks): wave lifetime mean %.0f ticks = %.2f us; first start -> last end %d ticks = %.2f us"
      % (life_wall.mean(), life_wall.mean() / 100.0, span_wall, span_wall / 100.0))
st[:, 46:48] = 0
w0 = wall[ok, 0] - wall[ok, 0].min(); w1 = wall[ok, 1] - wall[ok, 0].min()
pc = [0, 10, 25, 50, 75, 90, 100]
print("wave start (us) pct", pc, [round(float(np.percentile(w0, q)) / 100, 2) for q in pc])
print("wave end   (us) pct", pc, [round(float(np.percentile(w1, q)) / 100, 2) for q in pc])
print("waves with wall stamps %d; started within 1.5 us: %d; within 3 us: %d" % (ok.sum(), (w0 < 150).sum(), (w0 < 300).sum()))
hist, edges = np.histogram(w0 / 100.0, bins=np.arange(0, 40, 2.0))
print("start-time histogram (2 us bins):", hist.tolist())
hist, edges = np.histogram(w1 / 100.0, bins=np.arange(0, 40, 2.0))
print("end-time histogram   (2 us bins):", hist.tolist())
gw = np.nonzero(ok)[0]
for lo_, hi_ in ((0, 2048), (2048, 4096), (4096, 5120), (5120, 6400), (6400, 9000), (9000, 13000)):
    m = (gw >= lo_) & (gw < hi_)
    if not m.any(): continue
    print("  waves %4d-%4d: start mean %.2f us, end mean %.2f us" % (lo_, hi_, w0[m].mean() / 100, w1[m].mean() / 100))
print("waves stamped %d, proposals/wave %.2f" % (len(starts), len(setup) / max(1, len(starts))))
print("wave start offset   ", f(starts))
print("wave end offset     ", f(ends))
print("wave lifetime       ", f(np.array(ends) - np.array(starts)))
print("proposal setup      ", f(setup))
print("sweep (128 rows x2) ", f(sweep_t))
print("  first sweep       ", f(sweep_t[0::sweeps]))
print("  later sweeps      ", f([v for i, v in enumerate(sweep_t) if i % sweeps]))
print("reductions+store    ", f(red))
