"""csrc/bsr_span.h on the host (CPU): the structural claim "this candidate's column lies in the span of the chain's
current columns" must hold numerically whenever it is made -- on random trees of the real generator and on the
constructed cases it exists for (a negation moved, linear combinations of current trees)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bsr_oracle as O
from bsr.tape import NODE_DTYPE, flatten, pack
from conftest import node_from_spec, spec_from_node


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("span") / "libspan.so")
    src = os.path.join(ROOT, "tests", "native", "span_shim.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", out, src], check=True)
    L = C.CDLL(out)
    L.span_check.restype = C.c_int
    L.span_check.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return L


def _check(L, cur, cands, ks):
    rows, off = pack([flatten(t) for t in cur] + [flatten(t) for t in cands])
    ks = np.ascontiguousarray(ks, dtype=np.int32)
    out = np.zeros(len(cands), dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.int32)
    L.span_check(rows.ctypes.data, off.ctypes.data, len(cur), len(cands), ks.ctypes.data, out.ctypes.data)
    return out


def _col(tree, X):
    with np.errstate(all="ignore"):
        return O.allcal(O.tree_from_json(spec_from_node(tree)), pd.DataFrame(X))[:, 0]


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


def test_constructed_cases(shim):
    x = _leaf
    cur = [_un("cos", x(3)), _bi("+", x(4), _un("ln", x(5), 0.8234, 0.0783)), _bi("*", _un("cubic", x(5)), x(0)),
           _bi("+", x(1), x(1)), _un("cos", _un("square", x(6))), _un("neg", x(6)), _un("exp", x(1))]
    cands = [
        (3, _un("neg", x(1)), 1),                                   # -x1 next to x1 + x1
        (3, _bi("+", x(1), _un("neg", x(1))), 1),                   # the zero column
        (3, _bi("+", x(1), x(6)), 1),                               # (x1 + x1)/2 - (-x6)
        (0, _bi("+", _un("cos", x(3)), x(6)), 1),                   # cos(x3) - (-x6)
        (5, _un("neg", _bi("+", x(6), x(6))), 1),                   # 2 (-x6)
        (0, _un("cos", _un("neg", x(3))), 2),                       # cos is even
        (2, _bi("*", _un("cubic", _un("neg", x(5))), x(0)), 2),     # (-x5)^3 x0 = -(x5^3 x0)
        (2, _bi("*", _un("neg", _un("cubic", x(5))), _un("neg", x(0))), 2),
        (4, _un("cos", _un("square", _un("neg", x(6)))), 2),
        (6, _un("exp", _un("neg", x(1))), 0),                       # exp(-x1) is another column
        (1, _bi("+", x(4), _un("ln", x(5), 0.8234, 0.0784)), 0),    # another constant: the constant column is not in the span
        (1, _bi("+", _un("ln", x(5), 0.8234, 0.0783), x(4)), 2),    # operands of + order-free
        (0, _un("sin", x(3)), 0),
        (3, _bi("+", x(1), x(2)), 0),
        (5, _un("inv", _un("inv", x(6))), 1),                    # 1/(1/x6) is -(-x6), to rounding: in the span, not a bit-exact repeat
        (5, _un("inv", _un("neg", _un("inv", _un("neg", x(6))))), 1),
        (3, _bi("+", _un("inv", _un("inv", x(1))), x(6)), 1),       # x1 + x6 with an inverse undone
        (5, _un("inv", _un("inv", _un("inv", x(6)))), 0),          # 1/x6 is not in the span
        (3, _un("ln", x(2), 1e-12, 0.0), 0),                       # a tiny multiple of a column outside the span is outside
        (3, _un("ln", x(1), 1e-12, 0.0), 1),                       # ... of one inside, inside
    ]
    got = _check(shim, cur, [t for _, t, _ in cands], [k for k, _, _ in cands])
    assert got.tolist() == [w for _, _, w in cands]


def _rank_says_in_span(M, z):
    """The reference's own criterion (np.linalg.matrix_rank, codes/funcs.py:1226): appending z to the current columns
    does not raise the rank."""
    with np.errstate(all="ignore"):
        return np.linalg.matrix_rank(np.column_stack([M, z])) == np.linalg.matrix_rank(M)


def test_claims_on_random_trees_hold_numerically(shim):
    """Soundness on the real generator: whatever is claimed to be in the span is -- by the reference's own rank
    criterion at this N, and to a few roundings by least squares -- on random data, well scaled and ill scaled
    (features of magnitudes 1e-6 .. 1e6: a cancelled large term then leaves a residue far above the rank tolerance)."""
    rs = np.random.RandomState(11)
    N, d, K = 400, 6, 4
    np.random.seed(5)
    n_claims = 0
    for rep in range(60):
        X = rs.uniform(-2, 2, size=(N, d)) * np.pi   # (off the 2^-51 grid of uniform(), on which sums are exact)
        if rep % 2:
            X = X * 10.0 ** rs.uniform(-6, 6, size=d)
        trees = []
        while len(trees) < K + 40:
            root = O.ONode(0)
            O.grow(root, d, list(O.OPS), list(O.OP_WEIGHTS), list(O.OP_ARITY), -1, 1.0, 1.0)
            if O.count_nodes(root) < 25:
                trees.append(node_from_spec(spec_from_node(root)))
        cur, cands = trees[:K], trees[K:]
        # some candidates that are in the span by construction: sums and differences of current trees, negations
        cands += [_bi("+", node_from_spec(spec_from_node(cur[0])), node_from_spec(spec_from_node(cur[1]))),
                  _un("neg", node_from_spec(spec_from_node(cur[2]))),
                  _bi("+", _un("neg", node_from_spec(spec_from_node(cur[3]))), node_from_spec(spec_from_node(cur[0])))]
        ks = rs.randint(0, K, size=len(cands))
        got = _check(shim, cur, cands, ks)
        assert got[-2] >= 1            # a negated current tree is that tree up to sign, whatever it is
        M = np.stack([_col(t, X) for t in cur], 1)
        if not np.isfinite(M).all():
            continue
        for i in np.nonzero(got >= 1)[0]:
            z = _col(cands[i], X)
            if not np.isfinite(z).all() or np.abs(M).max() > 1e150 or np.abs(z).max() > 1e150:
                continue
            n_claims += 1
            assert _rank_says_in_span(M, z), (rep, spec_from_node(cands[i]))
            if got[i] == 2:   # a repeat up to sign: bit for bit
                c = M[:, ks[i]]
                assert np.array_equal(z, c) or np.array_equal(z, -c), (rep, spec_from_node(cands[i]))
    assert n_claims > 100


def test_a_cancelled_large_term_makes_no_claim(shim):
    """`(x2 + x1) + -x1` is x2 in algebra; in numbers it is x2 plus the rounding of x1, which for |x1| >> N |x2| the
    reference's rank gate sees as a column of its own (np.linalg.matrix_rank: full rank).  No claim may be made -- as a
    candidate, as a current tree, or inside a non-linear operator -- while `x1 + -x1`, one column minus itself, stays an
    exact zero."""
    x = _leaf
    rs = np.random.RandomState(3)
    N = 1000
    X = rs.uniform(-2, 2, size=(N, 4)) * np.pi
    X[:, 1] *= 1e9
    cancel = lambda: _bi("+", _bi("+", x(2), x(1)), _un("neg", x(1)))
    cur = [x(2), _un("sin", x(3)), x(0)]
    cands = [(1, cancel(), 0),
             (1, _un("cos", cancel()), 0),                       # ... nor cos of it next to cos(x2)
             (1, _bi("+", x(1), _un("neg", x(1))), 1),           # one column minus itself: exactly zero
             (1, _bi("+", x(2), x(2)), 1),
             (0, _bi("+", _un("ln", x(2), 1.0, 0.0), _un("ln", x(2), -0.999, 0.0)), 0)]   # 0.001 x2 with the rounding of x2
    got = _check(shim, cur, [t for _, t, _ in cands], [k for k, _, _ in cands])
    assert got.tolist() == [w for _, _, w in cands]
    M = np.stack([_col(t, X) for t in cur], 1)
    z = _col(cands[0][1], X)
    assert np.linalg.matrix_rank(np.column_stack([M[:, [0]], z])) == 2   # the case the claim would have got wrong
    # the same tree as a CURRENT tree: x2 is then not in the span of the basis by form alone
    cur2 = [cancel(), _un("sin", x(3)), x(0)]
    got2 = _check(shim, cur2, [x(2), cancel()], [1, 0])
    assert got2.tolist() == [0, 2]                                       # (itself again: a repeat, bit for bit)


def test_association_is_part_of_a_repeat(shim):
    """(x1 + x2) + x3 and x1 + (x2 + x3) differ in their last bits: in the span (to rounding), not a repeat (the fp32
    path relies on repeats being bit for bit); operands of one sum may trade places."""
    x = _leaf
    cur = [_bi("+", _bi("+", x(1), x(2)), x(3)), x(0), _un("ln", _un("ln", x(4), 10.0, 0.0), 0.1, 0.0)]
    cands = [(0, _bi("+", x(1), _bi("+", x(2), x(3))), 1),
             (0, _bi("+", x(3), _bi("+", x(2), x(1))), 2),
             (0, _un("neg", _bi("+", _bi("+", _un("neg", x(2)), _un("neg", x(1))), _un("neg", x(3)))), 2),
             (2, x(4), 1),                                               # 0.1 (10 x4) is x4 to rounding only
             (2, _un("ln", _un("ln", x(4), 10.0, 0.0), 0.1, 0.0), 2)]
    got = _check(shim, cur, [t for _, t, _ in cands], [k for k, _, _ in cands])
    assert got.tolist() == [w for _, _, w in cands]
    rs = np.random.RandomState(4)
    X = rs.uniform(-2, 2, size=(300, 5)) * np.pi     # (uniform() itself lies on a grid of 2^-51: its sums are exact)
    a, b, c = _col(cur[0], X), _col(cands[0][1], X), _col(cands[1][1], X)
    assert np.array_equal(a, c) and not np.array_equal(a, b)
