"""Direct AQL dispatch of scoring batches (csrc/bsr_aql.h): the library writes the batch's kernel-dispatch packets into
ROCr queues of its own instead of calling hipLaunchKernel three to five times.  Same kernels, same arguments, same
order -- so the same bytes as the HIP-stream path (BSR_AQL=0, a fresh process: the choice is per process and device), for
every row pass the contexts select, batches in flight on every slot, commits in between (a commit leaves work on the
slot's stream: the next batch of that slot goes through the stream), the native sampler's device-side MH step, and
timed batches (the packet processor's own timestamps instead of HIP events).  Needs an MI355X."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "mcmc-symreg_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from test_gpu_dispatch import run_case
out = {}
for name in json.loads(sys.argv[1]):
    res, info = run_case(name)
    np.save(os.path.join(sys.argv[2], name.replace(" ", "_") + ".npy"), res)
    out[name] = info
print(json.dumps(out))
"""

# name -> (N, d, K, dtype, max_batch, batches, depth)
CASES = {
    "c2 whole slices asm": (100_000, 10, 3, "f64", 64, 40, 6),
    "k8 whole slices": (60_000, 8, 8, "f64", 64, 12, 4),
    "streaming": (600_000, 12, 3, "f64", 32, 6, 3),
    "fp32 chunked": (50_000, 6, 2, "f32", 32, 10, 4),
    "tiny work queue": (700, 4, 1, "f64", 16, 12, 8),
}


def _trees(d, rs, n):
    from bsr.node import Node
    from bsr.tape import flatten

    def leaf(f):
        x = Node(1); x.type = 0; x.feature = np.array([f]); return x

    def un(op, c, a=None, b=None):
        x = Node(0); x.type, x.operator, x.left, x.a, x.b = 1, op, c, a, b; c.parent = x; return x

    def bi(op, l, r):
        x = Node(0); x.type, x.operator, x.left, x.right = 2, op, l, r; l.parent = r.parent = x; return x

    out = []
    for i in range(n):
        kind = i % 7
        a, b = leaf(int(rs.randint(d))), leaf(int(rs.randint(d)))
        if kind == 0: t = bi('*', a, b)
        elif kind == 1: t = un('sin', bi('+', a, b))
        elif kind == 2: t = un('ln', un('exp', a), float(rs.uniform(0.5, 1.5)), float(rs.uniform(-1, 1)))
        elif kind == 3: t = bi('+', un('cos', a), un('square', b))
        elif kind == 4: t = un('inv', un('ln', un('square', a), 1.0, 1.0))
        elif kind == 5: t = bi('*', bi('+', a, b), un('cubic', leaf(int(rs.randint(d)))))
        else: t = a
        out.append(flatten(t))
    return out


def _near_span(f):
    """ln(exp(x_f)): x_f up to rounding -- nearly inside the span of a chain that holds x_f, and not by construction
    (the canonical forms do not see through it): k_solve flags it for the residual pass."""
    from bsr.node import Node
    from bsr.tape import flatten
    x = Node(1); x.type = 0; x.feature = np.array([f])
    e = Node(0); e.type, e.operator, e.left = 1, 'exp', x; x.parent = e
    l = Node(0); l.type, l.operator, l.left, l.a, l.b = 1, 'ln', e, 1.0, 0.0; e.parent = l
    return flatten(l)


def run_case(name):
    """Scores the case's batches pipelined over `depth` slots, with a commit + refresh in the middle, and one timed
    batch; returns (all scores as one array, what the context says about its dispatch)."""
    from bsr import _lib
    from bsr.device import DeviceContext, pack
    N, d, K, dtype, B, n_batches, depth = CASES[name]
    rs = np.random.RandomState(11)
    X = rs.uniform(-2, 2, size=(N, d))
    y = X[:, 0] * X[:, 1] + np.sin(X[:, 2 % d]) + 0.1 * rs.standard_normal(N)
    ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=B, dtype=dtype)
    cur = _trees(d, rs, 7 * K)[::7][:K] if K <= 1 else _trees(d, rs, 7 * K)[:K]
    if K > 1:
        cur[0] = _trees(d, rs, 7)[6]         # a plain feature ...
        feat0 = int(cur[0]["feature"][0]) if "feature" in cur[0].dtype.names else 0
    for k in range(K):
        ctx.set_current(0, k, cur[k])
    ctx.refresh(0)
    batches = []
    for b in range(n_batches):
        tapes = _trees(d, rs, B)
        ks = rs.randint(K, size=B).astype(np.int32)
        if K > 1 and b % 3 == 0:             # ... and, every third batch, a candidate that all but repeats it on another tree
            tapes[1] = _near_span(feat0)
            ks[1] = 1
        rows, off = pack(tapes)
        batches.append((rows, off, np.zeros(B, np.int32), ks, rs.uniform(0.5, 1.5, size=B)))
    outs = [np.zeros(B, dtype=_lib.SCORE_DTYPE) for _ in batches]
    half = n_batches // 2
    for lo, hi in ((0, half), (half, n_batches)):
        tickets = []
        for i in range(lo, hi):
            tickets.append((ctx.score_submit(*batches[i]), i))
            if len(tickets) >= depth:
                t, j = tickets.pop(0)
                ctx.score_wait(t, outs[j])
        last = None
        while tickets:
            t, j = tickets.pop(0)
            ctx.score_wait(t, outs[j])
            last = (t, j)
        if lo == 0:
            # accept a full-rank candidate of the last waited batch, then go on: the slot's next batch finds work on its stream
            t, j = last
            ok = [i for i in range(B) if outs[j]["rank"][i] == K and np.isfinite(outs[j]["loglik"][i])]
            if ok:
                ctx.commit(0, int(batches[j][3][ok[0]]), ok[0])
                ctx.refresh(0)
    ctx.set_profiling(1)
    timed = ctx.score_packed(*batches[0], np.zeros(B, dtype=_lib.SCORE_DTYPE))
    us = float(ctx.last_timing()[0])
    ctx.set_profiling(0)
    info = ctx.dispatch_info()
    info["row_pass_us"] = us
    info["row_pass"] = ctx.info()["row_pass"]
    ctx.close()
    return np.concatenate(outs + [timed]), info


def _other_process(names, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, json.dumps(names), tmp], env=env,
                           capture_output=True, text=True, timeout=1200)
        assert p.returncode == 0, p.stderr[-3000:]
        info = json.loads(p.stdout.strip().splitlines()[-1])
        return {n: np.load(os.path.join(tmp, n.replace(" ", "_") + ".npy")) for n in names}, info


def test_direct_dispatch_scores_the_same_bytes_as_the_stream_path():
    names = list(CASES)
    streamed, info_s = _other_process(names, {"BSR_AQL": "0"})
    for n in names:
        assert not info_s[n]["direct"] and info_s[n]["batches_direct"] == 0, (n, info_s[n])
    kinds = set()
    for n in names:
        got, info = run_case(n)
        assert info["direct"] and info["queues"] == 4, (n, info)
        # all but the batch behind the commit (and perhaps the rescoring runs of K = 1) went out as packets
        assert info["batches_direct"] >= CASES[n][5] - 2, (n, info)
        if CASES[n][5] // 2 >= 8:    # the second half comes round to the slot the commit was made from: that batch
            assert info["batches_streamed"] >= 1, (n, info)      # finds the commit's work on its stream and takes the stream
        assert 0.5 < info["row_pass_us"] < 5e4, (n, info)         # the timed batch: the dispatch's own timestamps
        assert 0.5 < info_s[n]["row_pass_us"] < 5e4 and 0.25 < info["row_pass_us"] / info_s[n]["row_pass_us"] < 4.0, (n, info, info_s[n])
        assert got.tobytes() == streamed[n].tobytes(), n
        kinds.add(info["row_pass"])
    assert {"k_tile1a", "k_tile1", "k_stream"} <= kinds, kinds


def test_the_native_sampler_runs_the_same_chains_either_way():
    """Device-side MH (spans + k_events behind the batch) through packets and through the stream: same chains."""
    code = (
        "import os, sys, json, numpy as np\n"
        "sys.path.insert(0, os.path.join(%r, 'mcmc-symreg_amd'))\n"
        "from bsr.chain import DeviceScorer\n"
        "from bsr.native import NativeEngine\n"
        "from bsr.node import Express\n"
        "rs = np.random.RandomState(5)\n"
        "X = rs.uniform(-3, 3, size=(20000, 5)); y = 1.3 * X[:, 0] * X[:, 1] + np.sin(X[:, 2]) + 0.1 * rs.standard_normal(20000)\n"
        "sc = DeviceScorer(X, y, 2, n_chains=2, max_batch=64)\n"
        "eng = NativeEngine(sc.ctx, 2, 5, val=80)\n"
        "for c in range(2):\n"
        "    eng.seed(c, 300 + c); eng.init_chain(c)\n"
        "eng.run(batch_per_chain=16)\n"
        "res = [eng.result(c) for c in range(2)]\n"
        "info = sc.ctx.dispatch_info()\n"
        "eng.close(); sc.close()\n"
        "print(json.dumps({'chains': [[[Express(t) for t in r['roots']], int(r['n_props']), int(r['n_accept']), r['beta'].tobytes().hex(), [float(e) for e in r['errs']]] for r in res], 'info': info}))\n"
    ) % ROOT
    outs = {}
    for aql in ("1", "0"):
        env = dict(os.environ)
        env["BSR_AQL"] = aql
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        outs[aql] = json.loads(p.stdout.strip().splitlines()[-1])
    assert outs["1"]["info"]["direct"] and outs["1"]["info"]["batches_direct"] > 5, outs["1"]["info"]
    assert not outs["0"]["info"]["direct"]
    assert outs["1"]["chains"] == outs["0"]["chains"]


def test_the_stream_path_still_passes_the_api_sequences():
    """tests/test_gpu_ctx_sequence.py once more, in a process that dispatches through HIP streams only."""
    env = dict(os.environ)
    env["BSR_AQL"] = "0"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ctx_sequence.py"), "-x", "-q", "-m", "gpu"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]


def test_packets_wrap_around_the_queues_many_times():
    """Four rings of 1024 packets: 6000 pipelined batches of three packets each go round every ring four times, some of them
    straddling the end (sent as two reservations: csrc/bsr_aql.hip aql_submit).  Every batch must come back with the bytes
    its twin had the first time round."""
    from bsr import _lib
    from bsr.device import DeviceContext, pack
    rs = np.random.RandomState(3)
    N, d, K, B = 3000, 5, 2, 16
    X = rs.uniform(-2, 2, size=(N, d))
    y = X[:, 0] * X[:, 1] + 0.1 * rs.standard_normal(N)
    ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=B)
    try:
        cur = _trees(d, rs, 7)
        for k in range(K):
            ctx.set_current(0, k, cur[k])
        ctx.refresh(0)
        batches = []
        for b in range(7):
            rows, off = pack(_trees(d, rs, B))
            batches.append(ctx.prepare(rows, off, np.zeros(B, np.int32), rs.randint(K, size=B).astype(np.int32), rs.uniform(0.5, 1.5, size=B)))
        first = [None] * len(batches)
        outs = [np.zeros(B, dtype=_lib.SCORE_DTYPE) for _ in range(8)]
        tickets = []
        n_total, bad = 6000, 0
        for i in range(n_total + 8):
            if i < n_total:
                tickets.append((ctx.score_submit_prepared(batches[i % 7]), i))
            if len(tickets) >= 8 or (i >= n_total and tickets):
                t, j = tickets.pop(0)
                o = outs[j % 8]
                ctx.score_wait(t, o)
                if first[j % 7] is None:
                    first[j % 7] = o.tobytes()
                elif o.tobytes() != first[j % 7]:
                    bad += 1
        info = ctx.dispatch_info()
        assert info["direct"] and info["batches_direct"] >= n_total, info
        assert bad == 0, bad
    finally:
        ctx.close()


POISON_WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "mcmc-symreg_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from bsr import _lib
from bsr.device import DeviceContext
from bsr.tape import pack
from test_gpu_dispatch import _trees
rs = np.random.RandomState(3)
N, d, K = 400_000, 6, 3
X = rs.uniform(-3, 3, size=(N, d)); y = rs.standard_normal(N)
ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=64)
cur = _trees(d, rs, K)
for k in range(K):
    ctx.set_current(0, k, cur[k])
ctx.refresh(0)
assert ctx.dispatch_info()["direct"], "direct dispatch is off on this box"
tapes = _trees(d, rs, 64)
rows, off = pack(tapes)
ch = np.zeros(64, np.int32); ks = (np.arange(64) %% K).astype(np.int32); sg = np.full(64, 0.7)
out = np.zeros(64, dtype=_lib.SCORE_DTYPE)
t = ctx.score_submit(rows, off, ch, ks, sg)
try:
    ctx.score_wait(t, out)
    print("NO_TIMEOUT")
except _lib.BsrError as e:
    print("WAIT_ERROR", e.code)
try:
    ctx.score_submit(rows, off, ch, ks, sg)
    print("SUBMIT_ACCEPTED")
except _lib.BsrError as e:
    print("SUBMIT_REFUSED", e.code)
time.sleep(0.2)           # (the abandoned batch ends on its own; nothing it writes has been freed)
ctx.close()
print("CLOSED")
"""


def test_a_batch_that_never_completes_poisons_its_context():
    """when the wait for a directly dispatched batch gives up (queue error, a minute of silence) the packets
    may still be queued or running.  The context must then refuse further batches (the slot's buffers would be staged
    over under the GPU) and bsr_ctx_destroy must leave everything the GPU can reach allocated.  The give-up is forced
    with BSR_DEBUG_AQL_TIMEOUT_MS=-1 (the first unfinished poll counts as silence) in a process of its own."""
    r = subprocess.run([sys.executable, "-c", POISON_WORKER % {"root": ROOT}],
                       env=dict(os.environ, BSR_DEBUG_AQL_TIMEOUT_MS="-1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.split()
    assert "WAIT_ERROR" in lines and "SUBMIT_REFUSED" in lines and "CLOSED" in lines and "SUBMIT_ACCEPTED" not in lines, r.stdout
