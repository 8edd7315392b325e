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


def test_claims_on_random_trees_hold_numerically(shim):
    """Soundness on the real generator: whatever is claimed to be in the span is, to rounding, on random data."""
    rs = np.random.RandomState(11)
    N, d, K = 400, 6, 4
    X = rs.uniform(-2, 2, size=(N, d))
    np.random.seed(5)
    n_claims = 0
    for rep in range(60):
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
        assert (got[-3:] >= 1).all()
        M = np.stack([_col(t, X) for t in cur], 1)
        if not np.isfinite(M).all():
            continue
        for i in np.nonzero(got >= 1)[0]:
            z = _col(cands[i], X)
            if not np.isfinite(z).all() or np.abs(M).max() > 1e8 or np.abs(z).max() > 1e8:
                continue
            n_claims += 1
            coef = np.linalg.lstsq(M, z, rcond=None)[0]
            res = np.linalg.norm(z - M @ coef)
            assert res <= 1e-9 * max(1.0, np.linalg.norm(z)), (spec_from_node(cands[i]), res)
    assert n_claims > 100
