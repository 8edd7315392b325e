"""The whole-slice row pass with its tape loop in gfx950 assembly (csrc/bsr_tile_asm.hip: k_tile1a) against the compiler's
k_tile1 on the same batches: the same bytes in every score -- every operator of the block (zeros under inv, overflow in
the cube, huge / infinite sin and cos arguments that send the tape back to the C++ interpreter, two ln nodes, seventeen
entries, eight terminals), every K it is written for, slices of 3..8 blocks (passes of fewer than four blocks), tapes the
block does not take (stack machine, `log`, three ln nodes), and against the oracle.  Needs an MI355X."""
import numpy as np
import pandas as pd
import pytest

from conftest import node_from_spec, spec_from_node

pytestmark = pytest.mark.gpu

import bsr_oracle as O

ENV = ("BSR_TILE_ASM", "BSR_TILE_SPLIT", "BSR_TILE_LONG", "BSR_AUX_CUS", "BSR_TILE_T", "BSR_DERIVED")


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


def _chain(ops, d, rs):
    """A chain tape: a terminal, then unary operators and `(+|*) terminal` entries in the given order."""
    t = _leaf(int(rs.randint(d)))
    for op in ops:
        if op in ("+", "*"):
            t = _bi(op, t, _leaf(int(rs.randint(d))))
        elif op == "ln":
            t = _un("ln", t, float(rs.uniform(-1.5, 1.5)), float(rs.uniform(-1, 1)))
        else:
            t = _un(op, t)
    return t


def _special_tapes(d, rs):
    L = _leaf
    out = [
        L(0),                                                       # a leaf
        _un("inv", L(1)),                                           # zeros under inv (column 1 holds some)
        _un("cubic", _bi("*", L(2), L(2))),                         # overflow in the cube (column 2 holds 1e120)
        _un("sin", _bi("*", L(2), L(0))),                           # huge and infinite arguments: back to the C++ interpreter
        _un("cos", _un("exp", _bi("*", L(3), L(3)))),               # exp clipped at 200 -> 1e10: a huge cos argument
        _un("exp", _un("neg", _un("square", L(4)))),
        _un("ln", _un("ln", L(0), 1.25, -0.5), -0.75, 2.0),         # two ln nodes: both pairs of the program
        _un("ln", _un("ln", _un("ln", L(0), 1.25, -0.5), -0.75, 2.0), 0.5, 0.25),   # three: not the block's
        _bi("*", _bi("+", L(0), L(1)), _bi("+", L(2 % d), L(3 % d))),               # a stacked operand: the stack machine
        _chain(["+", "*", "+", "*", "+", "*", "+"], d, rs),         # eight terminals
        _chain(["+", "*", "+", "*", "+", "*", "+", "*"], d, rs),    # nine: not the block's
        _chain(["sin", "+", "cos", "*", "exp", "neg", "square", "+", "cubic", "inv", "ln", "*", "neg", "sin", "+", "ln"], d, rs),  # 17 entries
        _chain(["sin", "+", "cos", "*", "exp", "neg", "square", "+", "cubic", "inv", "ln", "*", "neg", "sin", "+", "ln", "neg"], d, rs),  # 18
        _un("sin", _un("sin", _un("sin", L(1)))),
        _un("inv", _un("inv", _un("neg", L(0)))),
        _un("square", _un("cubic", _un("cos", L(3 % d)))),
    ]
    return out


def _data(N, d, seed):
    rs = np.random.RandomState(seed)
    X = rs.uniform(-3, 3, size=(N, d))
    X[::97, 1] = 0.0                      # zeros (inv, div)
    X[5::1013, 2] = 1e120                 # overflow in cube / huge trigonometric arguments
    X[7::2029, 2] = -3e7
    y = X[:, 0] * X[:, 1] + np.sin(X[:, 3 % d]) + 0.1 * rs.standard_normal(N)
    return X, y


def _oracle_score(X, y, cur_trees, cand, k, sigma):
    with np.errstate(all="ignore"):
        df = pd.DataFrame(X)
        cols = [O.allcal(O.tree_from_json(spec_from_node(t)), df)[:, 0] for t in cur_trees]
        cols[k] = O.allcal(O.tree_from_json(spec_from_node(cand)), df)[:, 0]
        out = np.stack(cols, axis=1)
        if not np.isfinite(out).all():
            return None
        if np.linalg.matrix_rank(out) < len(cur_trees):
            return None
        return O.yloglike(y, out, sigma)


@pytest.mark.parametrize("N,d,K,env", [
    (100_000, 10, 3, {}),                                            # the headline geometry: 97 slices of 8 blocks
    (100_000, 10, 4, {}),
    (100_000, 10, 2, {}),
    (100_000, 10, 1, {}),
    (100_000, 10, 3, {"BSR_TILE_LONG": "0"}),                        # 192 slices of 4 blocks: one pass
    (130_000, 6, 3, {"BSR_TILE_LONG": "0"}),                         # 5 blocks: a pass of four and a pass of one
    (150_000, 6, 4, {"BSR_TILE_LONG": "0"}),                         # 6
    (180_000, 6, 2, {"BSR_TILE_LONG": "0"}),                         # 7
    (100_000, 10, 3, {"BSR_TILE_LONG": "0", "BSR_AUX_CUS": "0"}),    # 256 slices of 3 blocks: one short pass
    (100_000, 10, 3, {"BSR_TILE_LONG": "0", "BSR_TILE_T": "2"}),     # two tape groups
    (40_037, 5, 3, {}),                                              # a ragged row count: the block that holds row N is a leftover unit
])
def test_the_assembly_tape_loop_scores_the_same_bytes_as_the_compilers(N, d, K, env, monkeypatch):
    from bsr.tape import flatten
    X, y = _data(N, d, 11 + K)
    rs = np.random.RandomState(5)
    np.random.seed(41 + K)
    trees = []
    while len(trees) < K + 48:
        root = O.ONode(0)
        O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
        if O.count_nodes(root) < 30:
            trees.append(node_from_spec(spec_from_node(root)))
    cur = [_bi("*", _leaf(0), _leaf(1)), _un("sin", _leaf(3 % d)), _un("ln", _leaf(4 % d), 0.7, -0.2), _un("square", _leaf(0))][:K]
    cands = _special_tapes(d, rs) + trees[K:]
    cands = cands[:64]
    B = len(cands)
    tapes = [flatten(t) for t in cands]
    ks = (np.arange(B) % K).astype(np.int32)
    sig = rs.uniform(0.5, 2.0, size=B)
    zeros = np.zeros(B, np.int32)

    def run(extra):
        for k in ENV:
            monkeypatch.delenv(k, raising=False)
        for k, v in {**env, **extra}.items():
            monkeypatch.setenv(k, v)
        c = _ctx(X, y, K=K, n_chains=1, max_batch=64)
        info = c.info()
        for k in range(K):
            c.set_current(0, k, flatten(cur[k]))
        c.refresh(0)
        out = c.score_batch(tapes, zeros, ks, sig).copy()
        again = c.score_batch(tapes[::-1], zeros, ks[::-1].copy(), sig[::-1].copy()).copy()   # another cost order, other waves
        c.close()
        assert again[::-1].tobytes() == out.tobytes()
        return out, info

    base, info = run({})
    assert info["row_pass"] == "k_tile1a", info
    ref, info0 = run({"BSR_TILE_ASM": "0"})
    assert info0["row_pass"] == "k_tile1", info0
    assert (info0["row_slices"], info0["blocks_per_slice"]) == (info["row_slices"], info["blocks_per_slice"])
    bad = [i for i in range(B) if base[i].tobytes() != ref[i].tobytes()]
    assert not bad, (info, bad, [(base[i]["loglik"], ref[i]["loglik"]) for i in bad[:4]])
    assert run({"BSR_TILE_SPLIT": "0"})[0].tobytes() == base.tobytes()
    assert run({"BSR_DERIVED": "0"})[0].tobytes() == base.tobytes()
    # ... and against the oracle, where it scores the candidate (full rank, finite)
    n_checked = 0
    for i in range(B):
        if base["rank"][i] != K or not np.isfinite(base["loglik"][i]) or base["smin"][i] < 1e-5 * base["smax"][i]:
            continue
        want = _oracle_score(X, y, cur, cands[i], int(ks[i]), float(sig[i]))
        if want is None:
            continue
        assert abs(base["loglik"][i] - want) <= 1e-6 * max(1.0, abs(want)), (i, base["loglik"][i], want)
        n_checked += 1
    assert n_checked >= B // 4
