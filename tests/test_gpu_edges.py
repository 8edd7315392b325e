"""Edge cases of the HIP path through the C ABI: ragged and tiny row counts, deep trees (stack spill), the largest
tapes, error returns, and independence of a proposal's result from the batch it is scored in.  Needs an MI355X."""
import numpy as np
import pandas as pd
import pytest

from conftest import node_from_spec, spec_from_node

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

import bsr_oracle as O


def _ctx(*a, **k):
    from bsr.device import DeviceContext
    return DeviceContext(*a, **k)


def _leaf(f):
    from bsr.node import Node
    n = Node(1)
    n.type = 0
    n.feature = np.array([f])
    return n


def _un(op, c, a=None, b=None):
    from bsr.node import Node
    n = Node(0)
    n.type, n.operator, n.left, n.a, n.b = 1, op, c, a, b
    c.parent = n
    return n


def _bi(op, l, r):
    from bsr.node import Node
    n = Node(0)
    n.type, n.operator, n.left, n.right = 2, op, l, r
    l.parent = r.parent = n
    return n


def _oracle_col(tree, X):
    with np.errstate(all="ignore"):
        return O.allcal(O.tree_from_json(spec_from_node(tree)), pd.DataFrame(X))[:, 0]


@pytest.mark.parametrize("N", [1, 2, 3, 63, 64, 65, 127, 129, 255, 256, 257, 1023, 1025, 2049, 4097])
def test_ragged_and_tiny_row_counts(N):
    """Row counts around every tiling boundary (lane pair, wave sweep, row block, column padding): columns, max|z| and
    the scored quantities must not see the padding rows."""
    from bsr.tape import flatten
    rs = np.random.RandomState(N)
    d, K = 3, 2
    X = rs.uniform(-2, 2, size=(N, d))
    y = X[:, 0] * X[:, 1] + 0.3 * X[:, 2] + 0.05 * rs.standard_normal(N)
    ctx = _ctx(X, y, K=K, n_chains=1, max_batch=8)
    cur = [_bi("*", _leaf(0), _leaf(1)), _un("ln", _leaf(2), 0.7, -0.2)]
    for k in range(K):
        ctx.set_current(0, k, flatten(cur[k]))
    ctx.refresh(0)
    cands = [_bi("+", _leaf(0), _leaf(2)), _un("square", _leaf(1)), _un("inv", _un("ln", _leaf(0), 1.0, 3.5)),
             _bi("*", _un("neg", _leaf(2)), _bi("+", _leaf(0), _leaf(1))), _leaf(1)]
    tapes = [flatten(t) for t in cands]
    cols, maxabs, flags = ctx.eval_tapes(tapes)
    for i, t in enumerate(cands):
        want = _oracle_col(t, X)
        assert np.array_equal(cols[i], want), (N, i)            # exact opcodes only: bit for bit
        assert maxabs[i] == np.max(np.abs(want)) and flags[i] == 0
    ks = np.array([0, 1, 0, 1, 0], dtype=np.int32)
    sig = np.full(len(cands), 0.8)
    res = ctx.score_batch(tapes, np.zeros(len(cands), np.int32), ks, sig)
    cur_cols = np.stack([_oracle_col(t, X) for t in cur], axis=1)
    for i, t in enumerate(cands):
        want = O.score_proposal(cur_cols, int(ks[i]), _oracle_col(t, X), y, sig[i])
        assert (res["rank"][i] == K) == (want["rank"] == K), (N, i, res[i], want)
        if want["rank"] == K:
            assert abs(res["loglik"][i] - want["loglik"]) <= 1e-6 * abs(want["loglik"]), (N, i, res[i], want)
    ctx.close()


def _perfect(depth, op, next_feature, d):
    if depth == 0:
        return _leaf(next(next_feature) % d)
    return _bi(op, _perfect(depth - 1, op, next_feature, d), _perfect(depth - 1, op, next_feature, d))


@pytest.mark.parametrize("depth", [4, 5, 6, 8, 12])
def test_deep_trees_spill_the_register_stack(depth):
    """A perfect binary tree of depth D needs D+1 live values: beyond 4 the interpreter spills to its per-wave global
    area.  Depth 12 is BASELINE config 5's bound (8 191 nodes).  Sums of terminals are exact in any operand order."""
    import itertools
    from bsr.tape import flatten
    N, d = 700, 5
    rs = np.random.RandomState(depth)
    X = rs.randint(-8, 9, size=(N, d)).astype(np.float64)      # small integers: every partial sum is exact
    y = rs.standard_normal(N)
    ctx = _ctx(X, y, K=2, n_chains=1, max_batch=4)
    tree = _perfect(depth, "+", itertools.count(), d)
    tape = flatten(tree)
    assert len(tape) == 2 ** (depth + 1) - 1
    mixed = _bi("*", _perfect(min(depth, 6), "+", itertools.count(3), d), _un("ln", _perfect(3, "+", itertools.count(1), d), 0.5, 1.0))
    cols, maxabs, flags = ctx.eval_tapes([tape, flatten(mixed)])
    assert np.array_equal(cols[0], _oracle_col(tree, X))
    assert np.array_equal(cols[1], _oracle_col(mixed, X))
    # the same deep tree as a scored candidate (projection pass + solve) against the oracle
    cur = [_leaf(0), _un("square", _leaf(1))]
    for k in range(2):
        ctx.set_current(0, k, flatten(cur[k]))
    ctx.refresh(0)
    res = ctx.score_batch([tape], np.zeros(1, np.int32), np.array([1], np.int32), np.array([1.3]))
    cur_cols = np.stack([_oracle_col(t, X) for t in cur], axis=1)
    want = O.score_proposal(cur_cols, 1, _oracle_col(tree, X), y, 1.3)
    assert int(res["rank"][0]) == want["rank"] == 2
    assert abs(res["loglik"][0] - want["loglik"]) <= 1e-9 * abs(want["loglik"])
    ctx.close()


def test_error_returns_do_not_poison_the_context():
    """Malformed input is refused with the documented codes and the context keeps working afterwards."""
    from bsr import _lib
    from bsr.tape import NODE_DTYPE, flatten
    rs = np.random.RandomState(0)
    X = rs.uniform(-1, 1, size=(300, 2))
    y = rs.standard_normal(300)
    with pytest.raises((_lib.BsrError, ValueError)):
        _ctx(np.zeros((0, 2)), np.zeros(0), K=2, n_chains=1, max_batch=4)      # no rows
    ctx = _ctx(X, y, K=2, n_chains=1, max_batch=4)
    good = flatten(_bi("+", _leaf(0), _leaf(1)))
    bad_feature = good.copy()
    bad_feature["feature"][0] = 7
    with pytest.raises(_lib.BsrError) as e:
        ctx.eval_tapes([bad_feature])
    assert _lib.ERRORS[e.value.code] == "BSR_E_TAPE" and "feature" in str(e.value)
    dangling = good[:2].copy()                                                   # two values left on the stack
    with pytest.raises(_lib.BsrError):
        ctx.eval_tapes([dangling])
    unknown = good.copy()
    unknown["opcode"][2] = 11                                                    # stream-only code, not a node opcode
    with pytest.raises(_lib.BsrError):
        ctx.eval_tapes([unknown])
    too_long = np.zeros(16385 * 2 + 1, dtype=NODE_DTYPE)                         # > BSR_MAX_TAPE nodes
    with pytest.raises(_lib.BsrError):
        ctx.eval_tapes([too_long])
    with pytest.raises(_lib.BsrError):
        ctx.eval_tapes([good] * 5)                                               # more than max_batch
    with pytest.raises(_lib.BsrError):
        ctx.score_batch([good], np.zeros(1, np.int32), np.zeros(1, np.int32), np.ones(1))   # chain never set/refreshed
    cols, _, _ = ctx.eval_tapes([good])
    assert np.array_equal(cols[0], X[:, 0] + X[:, 1])
    ctx.close()


def test_a_proposal_scores_the_same_in_any_batch():
    """Bit-identical results whether a candidate is scored alone, in a small batch or in a full one, in any position,
    sync or pipelined: the native sampler's worker threads and the bench's batching rely on it."""
    from bsr.tape import flatten, pack
    N, d, K, B = 5000, 6, 3, 64
    rs = np.random.RandomState(11)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    ctx = _ctx(X, y, K=K, n_chains=1, max_batch=B)
    np.random.seed(5)
    trees = []
    while len(trees) < K + B:
        root = O.ONode(0)
        O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
        if O.count_nodes(root) < 60:
            trees.append(node_from_spec(spec_from_node(root)))
    for k in range(K):
        ctx.set_current(0, k, flatten(trees[k]))
    ctx.refresh(0)
    tapes = [flatten(t) for t in trees[K:]]
    ks = (np.arange(B) % K).astype(np.int32)
    sig = rs.uniform(0.5, 2.0, size=B)
    zeros = np.zeros(B, np.int32)
    full = ctx.score_batch(tapes, zeros, ks, sig).copy()
    for i in (0, 7, 31, 63):
        one = ctx.score_batch([tapes[i]], zeros[:1], ks[i:i + 1], sig[i:i + 1])
        assert one[0].tobytes() == full[i].tobytes(), i
    perm = rs.permutation(B)
    shuffled = ctx.score_batch([tapes[j] for j in perm], zeros, ks[perm], sig[perm])
    assert shuffled.tobytes() == full[perm].tobytes()
    # pipelined: four half batches in flight
    halves = [perm[:16], perm[16:32], perm[32:48], perm[48:]]
    tickets = []
    for h in halves:
        rows, off = pack([tapes[j] for j in h])
        tickets.append((ctx.score_submit(rows, off, zeros[:len(h)], ks[h], sig[h]), h, (rows, off)))
    for t, h, _keep in tickets:
        out = np.zeros(len(h), dtype=full.dtype)
        ctx.score_wait(t, out)
        assert out.tobytes() == full[h].tobytes()
    ctx.close()


def test_exact_opcode_trees_are_bit_identical_to_the_oracle():
    """300 random trees over the opcodes whose device arithmetic is exactly numpy's (terminal, inv, ln, neg, square, +, *):
    every column bit for bit.  Exercises the fused terminal+binary stream entries in all positions and stack depths."""
    from bsr.tape import flatten
    rs = np.random.RandomState(77)
    N, d = 257, 4
    X = rs.uniform(-2, 2, size=(N, d))
    X[::17, 1] = 0.0                                           # exact zeros: inv(0) = 0
    ctx = _ctx(X, None, K=1, n_chains=1, max_batch=64)

    def rand_tree(depth):
        r = rs.uniform()
        if depth == 0 or r < 0.25:
            return _leaf(rs.randint(d))
        if r < 0.6:
            op = ["inv", "ln", "neg", "square"][rs.randint(4)]
            return _un(op, rand_tree(depth - 1), rs.normal(1, 0.5), rs.normal(0, 0.5)) if op == "ln" else _un(op, rand_tree(depth - 1))
        return _bi("+*"[rs.randint(2)], rand_tree(depth - 1), rand_tree(depth - 1))
    trees = [rand_tree(rs.randint(1, 7)) for _ in range(300)]
    for lo in range(0, len(trees), 60):
        chunk = trees[lo:lo + 60]
        cols, maxabs, flags = ctx.eval_tapes([flatten(t) for t in chunk])
        for i, t in enumerate(chunk):
            want = _oracle_col(t, X)
            same = (cols[i] == want) | (np.isnan(cols[i]) & np.isnan(want))
            assert same.all(), (lo + i, int((~same).sum()))
            assert bool(flags[i] & 1) == bool(np.isinf(want).any()) and bool(flags[i] & 2) == bool(np.isnan(want).any())
    ctx.close()


def test_k1_rescored_candidates_keep_the_batch_order_for_commit():
    """treeNum = 1: a candidate whose |z|^2 leaves the double range is rescored with a matched prescale
    (BSR_F_SCALE_RETRY).  The rescoring run must not disturb the other candidates' records: committing index 0 of a
    batch whose index 3 was rescored adopts candidate 0's max|z| and flags, and the refreshed chain state is the
    oracle's (codes/funcs.py:1147-1174 on the single column)."""
    from bsr import _lib
    from bsr.tape import flatten
    N, d = 3000, 3
    rs = np.random.RandomState(4)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 2.0 * X[:, 1] + 0.3 * X[:, 0] ** 2 + 0.1 * rs.standard_normal(N)
    ctx = _ctx(X, y, K=1, n_chains=1, max_batch=8)
    ctx.set_current(0, 0, flatten(_un("sin", _leaf(0))))
    ctx.refresh(0)
    huge = _un("cubic", _un("exp", _un("ln", _leaf(0), 60.0, 0.0)))          # up to (e^180)^3 ~ 3e234 > 2^400
    cands = [_leaf(1), _un("square", _leaf(2)), _bi("+", _leaf(0), _leaf(1)), huge, _un("neg", _leaf(2))]
    res = ctx.score_batch([flatten(t) for t in cands], np.zeros(5, np.int32), np.zeros(5, np.int32), np.full(5, 0.9))
    assert not (res["flags"] & _lib.F_SCALE_RETRY).any()
    cols = [_oracle_col(t, X) for t in cands]
    for i, col in enumerate(cols):
        want = O.score_proposal(np.zeros((N, 1)), 0, col, y, 0.9)
        assert int(res["rank"][i]) == want["rank"] == 1, i
        assert abs(res["loglik"][i] - want["loglik"]) <= 1e-9 * abs(want["loglik"]), i
        assert res["maxabs"][i] == np.max(np.abs(col)), i
    assert res["maxabs"][3] > 2.0 ** 400
    ctx.commit(0, 0, 0)
    info = ctx.refresh(0)
    assert info["maxabs"][0] == np.max(np.abs(X[:, 1]))
    ll, sse, scale, beta = O.yloglike_parts(y, cols[0].reshape(-1, 1), 1.0)
    assert abs(info["sse_old"] - sse) <= 1e-9 * sse
    beta_i, rmse = ctx.fit_beta(0)
    want_b, want_rmse = O.intercept_fit(pd.Series(y), cols[0].reshape(-1, 1))
    assert np.allclose(beta_i.reshape(-1), np.asarray(want_b).reshape(-1), rtol=1e-8) and abs(rmse - want_rmse) <= 1e-9 * want_rmse
    # a commit that refers to a batch whose slot has been submitted to again is refused
    t = [ctx.score_submit(*__import__("bsr.tape", fromlist=["pack"]).pack([flatten(cands[0])]), np.zeros(1, np.int32),
                          np.zeros(1, np.int32), np.ones(1)) for _ in range(_lib.MAX_INFLIGHT)]
    out = np.zeros(1, dtype=_lib.SCORE_DTYPE)
    ctx.score_wait(t[0], out)
    t.append(ctx.score_submit(*__import__("bsr.tape", fromlist=["pack"]).pack([flatten(cands[1])]), np.zeros(1, np.int32),
                              np.zeros(1, np.int32), np.ones(1)))            # reuses the slot of t[0]
    with pytest.raises(_lib.BsrError) as e:
        ctx.commit(0, 0, 0)
    assert _lib.ERRORS[e.value.code] == "BSR_E_STATE"
    for tk in t[1:]:
        ctx.score_wait(tk, out)
    ctx.close()


@pytest.mark.parametrize("N", [130, 4097])
def test_work_queue_soak_against_the_static_grid(N, monkeypatch):
    """The ticket-queue launch of k_rows (per-XCD counters, four batches in flight, counters re-armed by the kernel
    itself) hammered with 10^4 short launches: every result bit-identical to the static-grid launch of the same
    kernel (LDS-staged variant), i.e. no ticket is ever skipped, even with rounds this short."""
    from bsr.tape import flatten, pack
    d, K, B = 4, 3, 24
    rs = np.random.RandomState(17)
    X = rs.uniform(-3, 3, size=(N, d))
    y = X[:, 0] * X[:, 1] + np.sin(X[:, 2]) + 0.1 * rs.standard_normal(N)
    np.random.seed(9)
    trees = []
    while len(trees) < K + B:
        root = O.ONode(0)
        O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
        if O.count_nodes(root) < 40:
            trees.append(node_from_spec(spec_from_node(root)))
    tapes = [flatten(t) for t in trees[K:]]
    ks = (np.arange(B) % K).astype(np.int32)
    sig = rs.uniform(0.5, 2.0, size=B)
    zeros = np.zeros(B, np.int32)

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = _ctx(X, y, K=K, n_chains=1, max_batch=B)
        for k in range(K):
            c.set_current(0, k, flatten(trees[k]))
        c.refresh(0)
        return c
    # the static grid with X staged in LDS (round 1's launch, BSR_NO_LDS=0) lives in the TEST library only
    # (csrc/build.sh variants): scored there in a child process, compared here with the shipped library's queue
    import subprocess, sys, tempfile
    variants = os.path.join(ROOT, "mcmc-symreg_amd", "bsr", "libbsr_hip_variants.so")
    assert os.path.exists(variants), "build the test library: bash mcmc-symreg_amd/csrc/build.sh variants"
    with tempfile.TemporaryDirectory() as tmp:
        rows0, off0 = pack(tapes)
        np.savez(os.path.join(tmp, "in.npz"), X=X, y=y, rows=rows0, off=off0, ks=ks, sig=sig,
                 cur_rows=pack([flatten(t) for t in trees[:K]])[0], cur_off=pack([flatten(t) for t in trees[:K]])[1])
        script = (
            "import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from bsr.device import DeviceContext\n"
            "z = np.load(sys.argv[1])\n"
            "K = len(z['cur_off']) - 1; B = len(z['ks'])\n"
            "c = DeviceContext(z['X'], z['y'], K=K, n_chains=1, max_batch=B)\n"
            "for k in range(K): c.set_current(0, k, z['cur_rows'][z['cur_off'][k]:z['cur_off'][k + 1]])\n"
            "c.refresh(0)\n"
            "out = np.zeros(B, dtype=__import__('bsr._lib', fromlist=['x']).SCORE_DTYPE)\n"
            "c.score_packed(z['rows'], z['off'], np.zeros(B, np.int32), z['ks'], z['sig'], out)\n"
            "np.save(sys.argv[2], np.frombuffer(out.tobytes(), dtype=np.uint8)); c.close()\n" % os.path.join(ROOT, "mcmc-symreg_amd"))
        env = dict(os.environ, BSR_LIB_PATH=variants, BSR_TILE="0", BSR_NO_LDS="0")
        p = subprocess.run([sys.executable, "-c", script, os.path.join(tmp, "in.npz"), os.path.join(tmp, "want.npy")],
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        want_bytes = np.load(os.path.join(tmp, "want.npy")).tobytes()
    queue = make({"BSR_TILE": "0", "BSR_NO_LDS": "1"})            # ticket queue
    rows, off = pack(tapes)
    from bsr import _lib
    out = np.zeros(B, dtype=_lib.SCORE_DTYPE)
    tickets = []
    n_launch = 10_000
    for i in range(n_launch):
        tickets.append(queue.score_submit(rows, off, zeros, ks, sig))
        if len(tickets) == 4:
            queue.score_wait(tickets.pop(0), out)
            assert out.tobytes() == want_bytes, i
    while tickets:
        queue.score_wait(tickets.pop(0), out)
        assert out.tobytes() == want_bytes
    queue.close()


@pytest.mark.parametrize("N,d,K", [(100_000, 10, 3), (40_000, 6, 8), (300_000, 40, 3)])
def test_tile_pass_variants_and_derived_columns_agree_to_the_bit(N, d, K, monkeypatch):
    """One batch through the tile pass with the slice staged whole and chunked through two LDS buffers (LDS-DMA), with and
    without derived columns, chain tapes through either evaluator: the per-lane sums grow block by block in row order in every variant and a derived column
    holds what the interpreter would compute inline, so the scores are bit-identical.  The work-queue row pass
    (different partial blocks) agrees to rounding."""
    import bsr_oracle as O
    from conftest import node_from_spec, spec_from_node
    from bsr.tape import flatten
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
    # a few candidates that repeat the tree they would replace (as it is, and negated at the root): recognised on the
    # host, scored without the residual pass (BSR_SELFDUP) -- to the bit what the residual pass makes of them
    from bsr.node import Node
    for j in range(min(K, 4)):
        tapes[j] = flatten(trees[j])
        ks[j] = j
    neg = Node(0)
    neg.type, neg.operator, neg.left = 1, 'neg', node_from_spec(spec_from_node(trees[0]))
    neg.left.parent = neg
    tapes[5] = flatten(neg)
    ks[5] = 0
    sig = rs.uniform(0.5, 2.0, size=B)
    zeros = np.zeros(B, np.int32)

    def run(env):
        for k in ("BSR_TILE", "BSR_TILE_CHUNK", "BSR_TILE_RING", "BSR_TILE_T", "BSR_DERIVED", "BSR_SUBMIT_THREAD",
                  "BSR_SELFDUP", "BSR_AUX_CUS", "BSR_BAR_WRITE", "BSR_CHAIN_EVAL", "BSR_REORDER", "BSR_TILE_ASM", "BSR_TILE_SPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = _ctx(X, y, K=K, n_chains=1, max_batch=B)
        for k in range(K):
            c.set_current(0, k, flatten(trees[k]))
        c.refresh(0)
        out = c.score_batch(tapes, zeros, ks, sig).copy()
        c.close()
        return out
    base = run({})
    # the tape loop in assembly (k_tile1a: the default for whole slices at K <= 4) against the compiler's k_tile1, and its
    # staging in two halves against everything at the first barrier
    assert run({"BSR_TILE_ASM": "0"}).tobytes() == base.tobytes()
    assert run({"BSR_TILE_SPLIT": "0"}).tobytes() == base.tobytes()
    # chain tapes through the stack machine (two blocks per pass) instead of the register-resident pass; operands of
    # + and * in tape order instead of fusing order: the same value for every row, the same sums
    assert run({"BSR_CHAIN_EVAL": "0"}).tobytes() == base.tobytes()
    assert run({"BSR_REORDER": "0"}).tobytes() == base.tobytes()
    assert run({"BSR_REORDER": "0", "BSR_CHAIN_EVAL": "0"}).tobytes() == base.tobytes()
    # the slice through two LDS buffers, a chunk of two / four blocks at a time (LDS-DMA), instead of staged whole
    assert run({"BSR_TILE_CHUNK": "2"}).tobytes() == base.tobytes()
    assert run({"BSR_TILE_CHUNK": "4"}).tobytes() == base.tobytes()
    assert run({"BSR_TILE_CHUNK": "2", "BSR_TILE_RING": "2"}).tobytes() == base.tobytes()   # one chunk ahead instead of three
    assert run({"BSR_TILE_CHUNK": "1", "BSR_CHAIN_EVAL": "0"}).tobytes() == base.tobytes()
    assert run({"BSR_SUBMIT_THREAD": "0"}).tobytes() == base.tobytes()   # the caller issues the HIP calls itself
    assert run({"BSR_BAR_WRITE": "0"}).tobytes() == base.tobytes()       # input block by hipMemcpyAsync, not host stores
    # a launch as wide as the machine: other slices, other partial sums -- the same scores to rounding
    wide = run({"BSR_AUX_CUS": "0"})
    okw = base["rank"] == K
    assert np.array_equal(wide["rank"], base["rank"])
    assert np.allclose(wide["loglik"][okw], base["loglik"][okw], rtol=1e-9, atol=0)
    assert run({"BSR_SELFDUP": "0"}).tobytes() == base.tobytes()         # self-duplicates through the residual pass
    # two tape groups (slices twice as long): other partial sums -- the same scores to rounding
    two = run({"BSR_TILE_T": "2"})
    assert np.array_equal(two["rank"], base["rank"])
    assert np.allclose(two["loglik"][okw], base["loglik"][okw], rtol=1e-9, atol=0)
    assert run({"BSR_DERIVED": "0"}).tobytes() == base.tobytes()
    assert run({"BSR_DERIVED": "0", "BSR_TILE_CHUNK": "2"}).tobytes() == base.tobytes()
    rows = run({"BSR_TILE": "0"})
    assert np.array_equal(rows["rank"], base["rank"])
    ok = base["rank"] == K
    assert np.allclose(rows["loglik"][ok], base["loglik"][ok], rtol=1e-9, atol=0)
    assert run({"BSR_TILE": "0", "BSR_DERIVED": "0"}).tobytes() == rows.tobytes()


@pytest.mark.parametrize("dtype", ["f64"])
def test_a_repeat_with_a_negation_moved_scores_like_the_repeat_it_is(dtype, monkeypatch):
    """Candidates that are the tree they would replace with a negation moved -- cos(-x), (-a) b, (-x)^3, 1/(-x),
    sin(-x), a - b written -b + a, a (-A) + b as (-a) A + b -- compute the old column up to sign, bit for bit: the host
    recognises them (canonical form with the signs carried to the root, csrc/bsr_span.h) and k_solve takes w = 0 without the
    residual step.  With the recognition off (BSR_SELFDUP=0) they go through the residual step: the same bytes."""
    from bsr.tape import flatten
    rs = np.random.RandomState(5)
    N, d, K = 30000, 5, 3
    X = rs.uniform(-2, 2, size=(N, d))
    X[:, 0] = rs.uniform(0.5, 2, size=N) * rs.choice([-1.0, 1.0], size=N)   # both signs, away from the pole of 1/sin
    y = np.sin(X[:, 0]) * X[:, 1] + X[:, 2] - 0.5 * X[:, 3] + 0.05 * rs.standard_normal(N)
    L = _leaf
    cur = [
        _bi("*", _un("cubic", _un("inv", _un("sin", L(0)))), _un("cos", L(1))),
        _bi("+", _un("ln", _un("square", L(2)), 0.7, -0.2), _un("neg", L(3))),
        _un("exp", _bi("*", L(4), L(0))),
    ]
    cands = [
        # tree 0: every odd function passes the sign on, cos and the product swallow it
        (0, _bi("*", _un("cubic", _un("inv", _un("sin", _un("neg", L(0))))), _un("cos", _un("neg", L(1))))),
        (0, _bi("*", _un("cos", L(1)), _un("neg", _un("cubic", _un("inv", _un("sin", _un("neg", L(0)))))))),
        (0, _un("neg", _bi("*", _un("cubic", _un("neg", _un("inv", _un("sin", L(0))))), _un("cos", L(1))))),
        # tree 1: a sum with the signs of both children flipped, operands swapped; ln of a negated operand
        (1, _un("neg", _bi("+", L(3), _un("neg", _un("ln", _un("square", _un("neg", L(2))), 0.7, -0.2))))),
        (1, _bi("+", _un("neg", L(3)), _un("ln", _un("neg", _un("square", L(2))), -0.7, -0.2))),
        # tree 2: exp keeps the sign of its operand inside: (-x4)(-x0) is x4 x0, (-x4) x0 is not
        (2, _un("exp", _bi("*", _un("neg", L(4)), _un("neg", L(0))))),
        (2, _un("exp", _bi("*", _un("neg", L(4)), L(0)))),
        (2, _un("exp", _bi("*", L(0), L(4)))),
    ]
    tapes = [flatten(t) for _, t in cands]
    ks = np.array([k for k, _ in cands], np.int32)
    sig = np.full(len(cands), 1.3)
    zeros = np.zeros(len(cands), np.int32)

    def run(env):
        for k in ("BSR_SELFDUP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = _ctx(X, y, K=K, n_chains=1, max_batch=16, dtype=dtype)
        for k in range(K):
            c.set_current(0, k, flatten(cur[k]))
        c.refresh(0)
        out = c.score_batch(tapes, zeros, ks, sig).copy()
        c.close()
        return out
    on = run({})
    off = run({"BSR_SELFDUP": "0"})
    assert on.tobytes() == off.tobytes()
    # the repeats score what the current state scores (the same model): all equal among themselves, full rank
    rep = [0, 1, 2, 3, 4, 5, 7]
    assert (on["rank"][rep] == K).all()
    assert np.ptp(on["sse"][rep]) <= 1e-9 * abs(on["sse"][0])
    # ... and exp(-x4 x0) is another model
    assert on["sse"][6] != on["sse"][0]
    # the oracle agrees on every one of them
    cur_cols = np.stack([_oracle_col(t, X) for t in cur], 1)
    for i, (k, t) in enumerate(cands):
        want = O.score_proposal(cur_cols, k, _oracle_col(t, X), y, sig[i])
        assert want["rank"] == on["rank"][i]
        assert np.isclose(on["loglik"][i], want["loglik"], rtol=1e-9, atol=0)


def test_a_linear_combination_of_current_trees_scores_the_same_with_and_without_the_shortcut(monkeypatch):
    """Candidates that are linear combinations of the chain's current trees (-x1 next to x1 + x1, x1 + x2 next to
    x1 + x1 and -x2, cos(x3) + x2, the zero column x1 + -x1) lie in the span by construction: the host says so
    (csrc/bsr_span.h) and k_solve takes w = 0 without the residual step.  With the recognition off they go through the
    residual step: the same bytes.  The oracle agrees on rank and log-likelihood."""
    from bsr.tape import flatten
    rs = np.random.RandomState(9)
    N, d, K = 30000, 5, 4
    X = rs.uniform(-2, 2, size=(N, d))
    y = X[:, 1] - 0.7 * X[:, 2] + np.cos(X[:, 3]) + 0.3 * X[:, 0] * X[:, 4] + 0.05 * rs.standard_normal(N)
    L = _leaf
    cur = [_bi("+", L(1), L(1)), _un("neg", L(2)), _un("cos", L(3)), _bi("*", L(0), L(4))]
    cands = [
        (0, _un("neg", L(1))),
        (0, _bi("+", L(1), _un("neg", L(1)))),
        (1, _bi("+", L(1), L(2))),
        (2, _bi("+", _un("cos", L(3)), L(2))),
        (1, _un("neg", _bi("+", L(2), L(2)))),
        (1, _un("inv", _un("inv", L(2)))),                      # x2 to rounding: in the span of -x2
        (3, _bi("+", _bi("*", L(4), L(0)), _un("ln", L(1), 2.5, 0.0))),
        (3, _bi("+", _bi("*", L(4), L(0)), _un("ln", L(1), 2.5, 0.3))),   # + a constant: not in the span
        (2, _un("sin", L(3))),
    ]
    tapes = [flatten(t) for _, t in cands]
    ks = np.array([k for k, _ in cands], np.int32)
    sig = np.full(len(cands), 0.9)
    zeros = np.zeros(len(cands), np.int32)

    def run(env):
        monkeypatch.delenv("BSR_SELFDUP", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = _ctx(X, y, K=K, n_chains=1, max_batch=16)
        for k in range(K):
            c.set_current(0, k, flatten(cur[k]))
        c.refresh(0)
        out = c.score_batch(tapes, zeros, ks, sig).copy()
        c.close()
        return out
    on = run({})
    off = run({"BSR_SELFDUP": "0"})
    assert on.tobytes() == off.tobytes()
    cur_cols = np.stack([_oracle_col(t, X) for t in cur], 1)
    for i, (k, t) in enumerate(cands):
        want = O.score_proposal(cur_cols, k, _oracle_col(t, X), y, sig[i])
        assert want["rank"] == on["rank"][i], (i, want["rank"], on["rank"][i])
        if want["rank"] == K:
            assert np.isclose(on["loglik"][i], want["loglik"], rtol=1e-9, atol=0)


def test_the_library_places_its_own_threads_and_leaves_the_caller_alone():
    """bsr_ctx_create places the library's own threads (submission threads) on one L3 domain of the host and does NOT
    touch the caller's affinity ("CPU placement"); BSR_PIN=1 confines the caller as well (bench.py asks for
    it), BSR_PIN=0 places nothing, BSR_PIN_CPUS names the CPUs.  Fresh interpreters: the choice is made once per process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, json, numpy as np\n"
        "sys.path.insert(0, os.path.join(%r, 'mcmc-symreg_amd'))\n"
        "from bsr.device import DeviceContext\n"
        "before = sorted(os.sched_getaffinity(0))\n"
        "X = np.random.RandomState(0).uniform(-1, 1, size=(300, 2))\n"
        "c = DeviceContext(X, X[:, 0], K=2, n_chains=1, max_batch=4)\n"
        "after = sorted(os.sched_getaffinity(0))\n"
        "me = os.getpid()\n"
        "others = []\n"
        "for t in os.listdir('/proc/self/task'):\n"
        "    if int(t) != me:\n"
        "        try: others.append(sorted(os.sched_getaffinity(int(t))))\n"
        "        except OSError: pass\n"
        "c.close()\n"
        "print(json.dumps({'before': before, 'after': after, 'others': others}))\n" % root)

    def run(extra):
        env = dict(os.environ)
        for k in ("BSR_PIN", "BSR_PIN_CPUS", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
            env.pop(k, None)
        env["BSR_SUBMIT_THREAD"] = "2"      # submission threads whatever the CPU budget
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    off = run({"BSR_PIN": "0"})
    assert off["after"] == off["before"]
    dflt = run({})
    assert dflt["after"] == dflt["before"]                     # the caller keeps its CPUs
    big = len(dflt["before"]) > 32 and os.path.exists("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % dflt["before"][0])
    if big:                                                    # ... a thread of the library sits on a strict subset
        assert any(len(o) < len(dflt["before"]) for o in dflt["others"])
    on = run({"BSR_PIN": "1"})
    assert set(on["after"]) <= set(on["before"]) and len(on["after"]) >= min(4, len(on["before"]))
    if big:
        assert len(on["after"]) < len(on["before"])
    if len(on["before"]) >= 8:
        pick = on["before"][:4]
        got = run({"BSR_PIN": "1", "BSR_PIN_CPUS": ",".join(str(c) for c in pick)})
        assert got["after"] == pick
    # two ranks on the node take different domains
    if big:
        r0 = run({"BSR_PIN": "1", "LOCAL_RANK": "0", "LOCAL_WORLD_SIZE": "2"})
        r1 = run({"BSR_PIN": "1", "LOCAL_RANK": "1", "LOCAL_WORLD_SIZE": "2"})
        assert not (set(r0["after"]) & set(r1["after"]))
