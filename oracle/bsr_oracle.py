"""CPU ORACLE for the MCMC-SymReg likelihood hot path.  TEST INFRASTRUCTURE ONLY.

This module is a plain Python/numpy restatement of the reference algorithm.  It
exists to CHECK the HIP path (tests/, __graft_entry__.smoke(), and bench.py's
cpu_baseline leg).  Nothing under mcmc-symreg_amd/ imports it, and the product
path must never route through it.

Parity status: PINNED.  Every function below is checked against golden vectors
produced by importing the real reference in the build container
(tools/gen_golden.py -> tests/golden/g1..g7; see tests/test_oracle_golden.py).
The reference ships no tests of its own.

Citations are file:line under /root/reference/.  Third-party arithmetic on the
path (numpy ufuncs / linalg, scipy.stats rvs/pdf, pandas Series.sum) is called
through the same libraries the reference calls, in the same order, because the
"accepted-tree sequence bit-exact" criterion pins the RNG draw order.

Two evaluation flavours:
  faithful=True   per-element Python loops for exp/inv like codes/funcs.py:184-195
  faithful=False  np.where for exp/inv (identical values, vectorised)
"""
import copy

import numpy as np
from scipy.stats import invgamma, norm

try:  # the reference indexes a DataFrame; the oracle accepts either
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None

OPS = ('inv', 'ln', 'neg', 'sin', 'cos', 'exp', 'square', 'cubic', '+', '*')  # codes/bsr_class.py:110
OP_ARITY = (1, 1, 1, 1, 1, 1, 1, 1, 2, 2)                                      # codes/bsr_class.py:112
OP_WEIGHTS = tuple([1.0 / len(OPS)] * len(OPS))                                # codes/bsr_class.py:111
LN = 'ln'
# Extensions beyond the reference's table.  The reference cannot evaluate them, so THIS file is where
# their semantics are defined: protected like the reference's inv (codes/funcs.py:189-195), element-wise numpy.
#   sub  x - y      div  where(y == 0, 0, x / y)      log  where(x == 0, 0, log|x|)   (natural logarithm)
EXT_OPS = ('sub', 'div', 'log')
EXT_ARITY = (2, 2, 1)


class ONode:
    """Tree node with the fields of codes/funcs.py:30-55."""

    def __init__(self, depth):
        self.type = -1
        self.order = 0
        self.left = None
        self.right = None
        self.depth = depth
        self.parent = None
        self.operator = None
        self.op_ind = None
        self.data = None
        self.feature = None
        self.a = None
        self.b = None


# ------------------------------------------------------------------ structure helpers
def preorder(node):
    """Node, left subtree, right subtree; refreshes .order (codes/funcs.py:127-142)."""
    acc = []
    stack = [node]
    while stack:
        n = stack.pop()
        acc.append(n)
        if n.left is not None:
            if n.right is not None:
                stack.append(n.right)
            stack.append(n.left)
    for i, n in enumerate(acc):
        n.order = i
    return acc


def height(node):  # codes/funcs.py:255-263
    if node.type == 0:
        return 0
    if node.type == 1:
        return 1 + height(node.left)
    return 1 + max(height(node.left), height(node.right))


def count_nodes(node):  # codes/funcs.py:269-277
    if node.type == 0:
        return 1
    if node.type == 1:
        return 1 + count_nodes(node.left)
    return 1 + count_nodes(node.left) + count_nodes(node.right)


def count_ln(node):  # codes/funcs.py:283-292
    if node.type == 0:
        return 0
    if node.type == 1:
        return (1 if node.operator == LN else 0) + count_ln(node.left)
    return count_ln(node.left) + count_ln(node.right)


def refresh_depth(node):  # codes/funcs.py:298-307
    node.depth = 0 if node.parent is None else node.parent.depth + 1
    if node.left is not None:
        refresh_depth(node.left)
        if node.right is not None:
            refresh_depth(node.right)


def express(node):  # codes/funcs.py:314-342
    if node.type == 0:
        return "x" + str(node.feature)
    if node.type == 1:
        inner = express(node.left)
        op = node.operator
        if op == 'exp':
            return "exp(" + inner + ")"
        if op == LN:
            return str(round(node.a, 4)) + "*(" + inner + ")+" + str(round(node.b, 4))
        if op == 'inv':
            return "1/[" + inner + "]"
        if op == 'sin':
            return "sin(" + inner + ")"
        if op == 'cos':
            return "cos(" + inner + ")"
        if op == 'square':
            return "(" + inner + ")^2"
        if op == 'cubic':
            return "(" + inner + ")^3"
        if op == 'log':
            return "log(" + inner + ")"
        return "-(" + inner + ")"
    if node.operator == '+':
        return express(node.left) + "+" + express(node.right)
    if node.operator == 'sub':
        return "(" + express(node.left) + ")-(" + express(node.right) + ")"
    if node.operator == 'div':
        return "(" + express(node.left) + ")/[" + express(node.right) + "]"
    return "(" + express(node.left) + ")*(" + express(node.right) + ")"


def tree_from_json(spec, parent=None):
    """Build an ONode tree from the plain-data form used by tests/golden/*.json."""
    if spec is None:
        return None
    n = ONode(spec["depth"])
    n.type = spec["type"]
    n.operator = spec["op"]
    n.op_ind = spec["op_ind"]
    n.feature = None if spec["feature"] is None else np.array([spec["feature"]])
    n.a = _unfnum(spec["a"])
    n.b = _unfnum(spec["b"])
    n.parent = parent
    n.left = tree_from_json(spec["left"], n)
    n.right = tree_from_json(spec["right"], n)
    return n


def tree_to_json(node):
    if node is None:
        return None
    feat = None if node.feature is None else int(np.asarray(node.feature).reshape(-1)[0])
    return {"type": int(node.type), "op": node.operator,
            "op_ind": None if node.op_ind is None else int(node.op_ind), "depth": int(node.depth),
            "feature": feat, "a": _fnum(node.a), "b": _fnum(node.b),
            "left": tree_to_json(node.left), "right": tree_to_json(node.right)}


def _unfnum(v):
    if isinstance(v, str):
        return float(v)
    return v


def _fnum(v):
    if v is None:
        return None
    v = float(v)
    if v != v:
        return "nan"
    if v == float("inf"):
        return "inf"
    if v == float("-inf"):
        return "-inf"
    return v


# ------------------------------------------------------------------ a-1: tree evaluation
def _column(indata, feature):
    if pd is not None and isinstance(indata, pd.DataFrame):
        return np.array(indata.iloc[:, feature])          # codes/funcs.py:178 -> fresh (N,1)
    return np.array(np.asarray(indata)[:, np.asarray(feature).reshape(-1)])


def allcal(node, indata, faithful=False):
    """Post-order evaluation over all rows -> (N,1) float array (codes/funcs.py:175-220)."""
    t = node.type
    if t == 0:
        node.data = _column(indata, node.feature)
    elif t == 1:
        v = allcal(node.left, indata, faithful)
        op = node.operator
        if op == LN:
            node.data = node.a * v + node.b                               # :181 two roundings
        elif op == 'exp':
            if faithful:                                                   # :182-188, same element access pattern
                for i in np.arange(len(v[:, 0])):                          # (2-D indexing per element: it IS the cost)
                    if v[i, 0] <= 200:
                        v[i, 0] = np.exp(v[i, 0])
                    else:
                        v[i, 0] = 1e+10
                node.data = v
            else:
                with np.errstate(all="ignore"):
                    node.data = np.where(v <= 200, np.exp(np.where(v <= 200, v, 0.0)), 1e+10)
        elif op == 'inv':
            if faithful:                                                   # :189-195
                for i in np.arange(len(v[:, 0])):
                    if v[i, 0] == 0:
                        v[i, 0] = 0
                    else:
                        v[i, 0] = 1 / v[i, 0]
                node.data = v
            else:
                with np.errstate(all="ignore"):
                    node.data = np.where(v == 0, 0.0, 1.0 / np.where(v == 0, 1.0, v))
        elif op == 'neg':
            node.data = -1 * v                                             # :197
        elif op == 'sin':
            node.data = np.sin(v)                                          # :199
        elif op == 'cos':
            node.data = np.cos(v)                                          # :201
        elif op == 'square':
            node.data = np.square(v)                                       # :203
        elif op == 'cubic':
            node.data = np.power(v, 3)                                     # :205
        elif op == 'log':                                                  # extension (EXT_OPS)
            with np.errstate(all="ignore"):
                node.data = np.where(v == 0, 0.0, np.log(np.abs(np.where(v == 0, 1.0, v))))
        else:
            raise ValueError("no matching unary operator %r" % (op,))
    elif t == 2:
        lv = allcal(node.left, indata, faithful)
        rv = allcal(node.right, indata, faithful)
        if node.operator == '+':
            node.data = lv + rv                                            # :210
        elif node.operator == '*':
            node.data = lv * rv                                            # :212
        elif node.operator == 'sub':                                       # extensions (EXT_OPS)
            node.data = lv - rv
        elif node.operator == 'div':
            with np.errstate(all="ignore"):
                node.data = np.where(rv == 0, 0.0, lv / np.where(rv == 0, 1.0, rv))
        else:
            raise ValueError("no matching binary operator %r" % (node.operator,))
    else:
        raise ValueError("not a grown tree (type %r)" % (t,))
    return node.data


# ------------------------------------------------------------------ a-2: OLS + Gaussian log-likelihood
def yloglike_parts(y, outputs, sigma):
    """Returns (loglik, sse, scale, beta) following codes/funcs.py:1147-1174 line by line."""
    XX = copy.deepcopy(outputs)
    scale = np.max(np.abs(XX))
    XX = XX / scale
    ridge = np.eye(XX.shape[1]) * 1e-6
    yy = np.array(y)
    yy.shape = (yy.shape[0], 1)
    beta = np.linalg.inv(np.matmul(XX.transpose(), XX) + ridge)
    beta = np.matmul(beta, np.matmul(XX.transpose(), yy))
    fitted = np.matmul(XX, beta)
    # y may be a pandas Series: np.sum then dispatches to Series.sum(skipna=True) (SURVEY hard part 8)
    sse = np.sum(np.square(y - fitted[:, 0]))
    ll = -sse / (2 * sigma * sigma)
    ll -= 0.5 * len(y) * np.log(2 * np.pi * sigma * sigma)
    return ll, sse, scale, beta


def yloglike(y, outputs, sigma):
    return yloglike_parts(y, outputs, sigma)[0]


def intercept_fit(y, cols):
    """[1|cols] ridge OLS of codes/bsr_class.py:147-163 / 211-233 -> (Beta (K+1,1), rmse)."""
    n = cols.shape[0]
    XX = np.concatenate((np.ones((n, 1)), cols), axis=1)
    scale = np.max(np.abs(XX))
    XX = XX / scale
    ridge = np.eye(XX.shape[1]) * 1e-6
    yy = np.array(y)
    yy.shape = (yy.shape[0], 1)
    beta = np.linalg.inv(np.matmul(XX.transpose(), XX) + ridge)
    beta = np.matmul(beta, np.matmul(XX.transpose(), yy))
    fitted = np.matmul(XX, beta)
    beta = beta / scale
    yv = np.asarray(y, dtype=np.float64)
    err = 0
    for i in range(n):                                                     # :229-231 sequential sum
        err += (fitted[i, 0] - yv[i]) * (fitted[i, 0] - yv[i])
    return beta, np.sqrt(err / n)


# ------------------------------------------------------------------ a-H: host driver (RNG order pinned)
def grow(node, nfeature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b):
    """codes/funcs.py:74-119."""
    depth = node.depth
    pick_op = True
    if depth > 0:
        prob = 1 / np.power((1 + depth), -beta)
        u = np.random.uniform(0, 1, 1)
        if u > prob:
            node.feature = np.random.randint(0, nfeature, size=1)         # :83 (overwritten at :99)
            node.type = 0
            pick_op = False
    if pick_op:
        k = np.random.choice(np.arange(len(Ops)), p=Op_weights)
        node.operator = Ops[k]
        node.type = Op_type[k]
        node.op_ind = k
    if node.type == 0:
        node.feature = np.random.randint(0, nfeature, size=1)             # :99
    elif node.type == 1:
        node.left = ONode(depth + 1)
        node.left.parent = node
        if node.operator == LN:
            node.a = norm.rvs(loc=1, scale=np.sqrt(sigma_a))
            node.b = norm.rvs(loc=0, scale=np.sqrt(sigma_b))
        grow(node.left, nfeature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b)
    else:
        node.left = ONode(depth + 1)
        node.left.parent = node
        node.right = ONode(depth + 1)
        node.right.parent = node
        grow(node.left, nfeature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b)
        grow(node.right, nfeature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b)


def fstruc(node, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b):
    """[log prior of structure, log prior of ln parameters]  (codes/funcs.py:349-398)."""
    ls = 0
    lp = 0
    if node.type == 0:
        ls += np.log(1 - 1 / np.power((1 + node.depth), -beta))
        ls -= np.log(n_feature)
    else:
        if node.depth == 0:
            ls += np.log(Op_weights[node.op_ind])
        else:
            ls += np.log((1 + node.depth)) * beta + np.log(Op_weights[node.op_ind])
        if node.type == 1 and node.operator == LN:
            lp -= np.power((node.a - 1), 2) / (2 * sigma_a)
            lp -= np.power(node.b, 2) / (2 * sigma_b)
            lp -= 0.5 * np.log(2 * np.pi * sigma_a)
            lp -= 0.5 * np.log(2 * np.pi * sigma_b)
    if node.left is None:
        return [ls, lp]
    sub = fstruc(node.left, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b)
    ls += sub[0]
    lp += sub[1]
    if node.right is None:
        return [ls, lp]
    sub = fstruc(node.right, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b)
    ls += sub[0]
    lp += sub[1]
    return [ls, lp]


def _detransform_candidates(tree):
    """Non-terminals, minus a root whose children are all terminal (codes/funcs.py:454-468)."""
    out = []
    for n in tree:
        ok = n.type != 0
        if n.parent is None:
            if n.right is None and n.left.type == 0:
                ok = False
            elif n.left.type == 0 and n.right.type == 0:
                ok = False
        if ok:
            out.append(n)
    return out


def _split_terms(tree):
    term = [n for n in tree if n.type == 0]
    nterm = [n for n in tree if n.type != 0]
    return term, nterm


def _replace_child(parent, old, new):
    if parent.left is old:
        parent.left = new
    else:
        parent.right = new
    new.parent = parent


def prop(Root, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b, info=None):
    """One structural proposal (codes/funcs.py:406-923).

    Returns [oldRoot, Root, lnPointers, change, Q, Qinv, last_a, last_b, cnode]."""
    G = (n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b)
    oldRoot = copy.deepcopy(Root)
    Tree = preorder(Root)
    lnPointers = [n for n in Tree if n.operator == LN]
    last_a = [n.a for n in lnPointers]
    last_b = [n.b for n in lnPointers]
    Term, Nterm = _split_terms(Tree)
    ltNum = len(lnPointers)
    cnode = None
    change = ''
    Q = Qinv = 1
    detcd = _detransform_candidates(Tree)

    p_stay = 0.25 * ltNum / (ltNum + 3)                                    # :475-480
    p_grow = (1 - p_stay) * min(1, 4 / (len(Nterm) + 2)) / 3
    p_prune = (1 - p_stay) / 3 - p_grow
    p_detr = (1 - p_stay) * (1 / 3) * len(detcd) / (3 + len(detcd))
    p_trans = (1 - p_stay) / 3 - p_detr
    p_rop = (1 - p_stay) / 6

    u = np.random.uniform(0, 1, 1)[0]                                      # :483

    if u <= p_stay:                                                        # :490-500
        action = 'stay'
        Q = p_stay
        Qinv = p_stay
        for n in Tree:
            if n.operator == LN:
                n.a = norm.rvs(loc=1, scale=np.sqrt(sigma_a))
                n.b = norm.rvs(loc=1, scale=np.sqrt(sigma_b))

    elif u <= p_stay + p_grow:                                             # :503-536
        action = 'grow'
        pod = np.random.randint(0, len(Term), 1)[0]
        tgt = Term[pod]
        grow(tgt, *G)
        if tgt.type == 0:
            Q = Qinv = 1
        else:
            fs = fstruc(tgt, *G)
            Q = p_grow * np.exp(fs[0]) / len(Term)
            new_lt = count_ln(Root)
            new_n = count_nodes(Root)
            nt, nnt = _split_terms(preorder(Root))
            new_p = (1 - 0.25 * new_lt / (new_lt + 3)) * (1 - min(1, 4 / (len(nnt) + 2))) / 3
            Qinv = new_p / max(1, (new_n - len(nt) - 1))
            if new_lt > ltNum:
                change = 'expansion'

    elif u <= p_stay + p_grow + p_prune:                                   # :539-579
        action = 'prune'
        pod = np.random.randint(1, len(Nterm), 1)[0]
        tgt = Nterm[pod]
        fs = fstruc(tgt, *G)
        if count_ln(tgt) > 0:
            change = 'shrinkage'
        tgt.left = None
        tgt.right = None
        tgt.operator = None
        tgt.type = 0
        tgt.feature = np.random.randint(0, n_feature, 1)
        new_lt = count_ln(Root)
        nt, nnt = _split_terms(preorder(Root))
        Q = p_prune / ((len(Nterm) - 1) * n_feature)
        pg = 1 - 0.25 * new_lt / (new_lt + 3) * 0.75 * min(1, 4 / (len(nnt) + 2))
        Qinv = pg * np.exp(fs[0]) / len(nt)

    elif u <= p_stay + p_grow + p_prune + p_detr:                          # :582-673
        action = 'detransform'
        det_od = np.random.randint(0, len(detcd), 1)[0]
        dn = detcd[det_od]
        cutt = None
        Q = p_detr / len(detcd)
        if dn.parent is None:
            if dn.right is None:
                Root = Root.left
            else:
                if dn.left.type == 0:
                    cutt = Root.left
                    Root = Root.right
                elif dn.right.type == 0:
                    cutt = Root.right
                    Root = Root.left
                else:
                    aa = np.random.uniform(0, 1, 1)[0]
                    if aa <= 0.5:
                        cutt = Root.right
                        Root = Root.left
                    else:
                        cutt = Root.left
                        Root = Root.right
                    Q = Q / 2
            Root.parent = None
            refresh_depth(Root)
        else:
            if dn.type == 1:
                _replace_child(dn.parent, dn, dn.left)
            else:
                aa = np.random.uniform(0, 1, 1)[0]
                if aa <= 0.5:
                    cutt = dn.right
                    _replace_child(dn.parent, dn, dn.left)
                else:
                    cutt = dn.left
                    _replace_child(dn.parent, dn, dn.right)
                Q = Q / 2
            Root.parent = None
            refresh_depth(Root)
        new_tree = preorder(Root)
        new_lt = sum(1 for n in new_tree if n.operator == LN)
        if new_lt < ltNum:
            change = 'shrinkage'
        new_pstay = 0.25 * new_lt / (new_lt + 3)
        new_detcd = _detransform_candidates(new_tree)
        new_pdetr = (1 - new_pstay) * (1 / 3) * len(new_detcd) / (len(new_detcd) + 3)
        new_ptr = (1 - new_pstay) / 3 - new_pdetr
        Qinv = new_ptr * Op_weights[dn.op_ind] / len(new_tree)
        if cutt is not None:
            fs = fstruc(cutt, *G)
            Qinv = Qinv * np.exp(fs[0])

    elif u <= p_stay + p_grow + p_prune + p_detr + p_trans:                # :679-786
        action = 'transform'
        Tree = preorder(Root)
        ins_ind = np.random.randint(0, len(Tree), 1)[0]
        ins = Tree[ins_ind]
        k = np.random.choice(np.arange(0, len(Ops)), p=Op_weights)
        w = Op_weights[k]
        nn = ONode(ins.depth)
        nn.operator = Ops[k]
        nn.type = Op_type[k]
        nn.op_ind = k
        if nn.type == 1 and nn.operator == LN:
            change = 'expansion'
        if ins.parent is None:
            Root = nn
        else:
            par = ins.parent
            if par.left is ins:
                par.left = nn
            else:
                par.right = nn
            nn.parent = par
        nn.left = ins
        ins.parent = nn
        if nn.type == 1:
            refresh_depth(Root)
            Q = p_trans * w / len(Tree)
        else:
            nr = ONode(1 if nn.parent is None else nn.depth + 1)
            nn.right = nr
            nr.parent = nn
            refresh_depth(Root)
            grow(nr, *G)
            fs = fstruc(nr, *G)
            Q = p_trans * w * np.exp(fs[0]) / len(Tree)
        new_tree = preorder(Root)
        new_lt = sum(1 for n in new_tree if n.operator == LN)
        if new_lt > ltNum:
            change = 'expansion'
        new_pstay = 0.25 * new_lt / (new_lt + 3)
        new_detcd = _detransform_candidates(new_tree)
        new_pdetr = (1 - new_pstay) * (1 / 3) * len(new_detcd) / (len(new_detcd) + 3)
        Qinv = new_pdetr / len(new_detcd)
        if nn.type == 2 and nn.left.type > 0 and nn.right.type > 0:
            Qinv = Qinv / 2

    elif u <= p_stay + p_grow + p_prune + p_detr + p_trans + p_rop:        # :791-903
        action = 'ReassignOperator'
        pod = np.random.randint(0, len(Nterm), 1)[0]
        cnode = Nterm[pod]
        last_op = cnode.operator
        last_oi = cnode.op_ind
        last_type = cnode.type
        k = np.random.choice(np.arange(0, len(Ops)), p=Op_weights)
        new_op = Ops[k]
        new_type = Op_type[k]
        if last_type == 1 and new_type == 1:
            cnode.operator = new_op
            if last_op == LN:
                if new_op != LN:
                    cnode.a = None
                    cnode.b = None
                    change = 'shrinkage'
            elif new_op == LN:
                change = 'expansion'
            Q = Op_weights[k]
            Qinv = Op_weights[last_oi]
        elif last_type == 1:
            cnode.operator = new_op
            cnode.type = 2
            if last_op == LN:
                cnode.a = None
                cnode.b = None
            cnode.right = ONode(cnode.depth + 1)
            cnode.right.parent = cnode
            grow(cnode.right, *G)
            fs = fstruc(cnode.right, *G)
            Q = p_rop * np.exp(fs[0]) * Op_weights[k] / (len(Nterm))
            new_n = count_nodes(Root)
            nt, _ = _split_terms(preorder(Root))
            new_lt = count_ln(Root)
            new_p0 = new_lt / (4 * (new_lt + 3))
            Qinv = 0.125 * (1 - new_p0) * Op_weights[last_oi] / (new_n - len(nt))
            if new_lt > ltNum:
                change = 'expansion'
            elif new_lt < ltNum:
                change = 'shrinkage'
        elif new_type == 1:
            cutted = copy.deepcopy(cnode.right)
            p_lt = count_ln(cutted)
            if p_lt > 1:
                change = 'shrinkage'
            elif new_op == LN:
                if p_lt == 0:
                    change = 'expansion'
            cnode.right = None
            cnode.operator = new_op
            cnode.type = new_type
            Q = p_rop * Op_weights[k] / len(Nterm)
            new_n = count_nodes(Root)
            preorder(Root)
            new_lt = count_ln(Root)
            new_p0 = new_lt / (4 * (new_lt + 3))
            fs = fstruc(cutted, *G)
            Qinv = 0.125 * (1 - new_p0) * np.exp(fs[0]) * Op_weights[last_oi] / (new_n - 0)   # :893-894
        else:
            cnode.operator = new_op
            Q = Op_weights[k]
            Qinv = Op_weights[last_oi]

    else:                                                                  # :907-917
        action = 'ReassignFeature'
        pod = np.random.randint(0, len(Term), 1)[0]
        fod = np.random.randint(0, n_feature, 1)
        Term[pod].feature = fod
        Q = Qinv = 1

    Root.parent = None
    refresh_depth(Root)
    if info is not None:
        info["action"] = action
    return [oldRoot, Root, lnPointers, change, Q, Qinv, last_a, last_b, cnode]


def auxprop(change, oldRoot, Root, lnPointers, sigma_a, sigma_b, last_a, last_b, cnode=None):
    """Auxiliary-variable step for ln parameters (codes/funcs.py:935-1138)."""
    Tree = preorder(Root)
    odList = [i for i, n in enumerate(Tree) if n.operator == LN]
    new_sa2 = invgamma.rvs(1)
    new_sb2 = invgamma.rvs(1)
    old_sa2 = sigma_a
    old_sb2 = sigma_b

    if change == 'shrinkage':
        keep_a, keep_b, cut_a, cut_b = [], [], [], []
        for i, p in enumerate(lnPointers):
            if p.operator == LN:
                keep_a.append(last_a[i])
                keep_b.append(last_b[i])
            else:
                cut_a.append(last_a[i])
                cut_b.append(last_b[i])
        for i in range(len(odList) - len(keep_a)):
            keep_a.append(cut_a[i])
            keep_b.append(cut_b[i])
        n0 = len(keep_a)
        Ua, Ub = [], []
        for _ in range(n0):
            Ua.append(norm.rvs(loc=0, scale=np.sqrt(new_sa2)))
            Ub.append(norm.rvs(loc=0, scale=np.sqrt(new_sb2)))
        Na = [keep_a[i] + Ua[i] for i in range(n0)]
        Nb = [keep_b[i] + Ub[i] for i in range(n0)]
        NUa = [keep_a[i] - Ua[i] for i in range(n0)] + last_a
        NUb = [keep_b[i] - Ub[i] for i in range(n0)] + last_b
        logh = 0
        loghstar = 0
        logh += np.log(invgamma.pdf(new_sa2, 1))
        logh += np.log(invgamma.pdf(new_sb2, 1))
        loghstar += np.log(invgamma.pdf(old_sa2, 1))
        loghstar += np.log(invgamma.pdf(old_sb2, 1))
        for i in range(len(Ua)):
            logh += np.log(norm.pdf(Ua[i], loc=0, scale=np.sqrt(new_sa2)))
            logh += np.log(norm.pdf(Ub[i], loc=0, scale=np.sqrt(new_sb2)))
        for i in range(len(NUa)):
            loghstar += np.log(norm.pdf(NUa[i], loc=0, scale=np.sqrt(old_sa2)))
            loghstar += np.log(norm.pdf(NUb[i], loc=0, scale=np.sqrt(old_sb2)))
        hratio = np.exp(loghstar - logh)
        detjacob = np.power(2, 2 * len(keep_a))
        for i in range(len(odList)):
            Tree[odList[i]].a = Na[i]
            Tree[odList[i]].b = Nb[i]
        return [hratio, detjacob, new_sa2, new_sb2]

    if change == 'expansion':
        new_sa2 = invgamma.rvs(1)
        new_sb2 = invgamma.rvs(1)
        m = len(last_a)
        Ua, Ub = [], []
        for _ in range(m):
            Ua.append(norm.rvs(loc=0, scale=np.sqrt(new_sa2)))
            Ub.append(norm.rvs(loc=0, scale=np.sqrt(new_sb2)))
        Na = [(last_a[i] + Ua[i]) / 2 for i in range(m)]
        Nb = [(last_b[i] + Ub[i]) / 2 for i in range(m)]
        NUa = [(last_a[i] - Ua[i]) / 2 for i in range(m)]
        NUb = [(last_b[i] - Ub[i]) / 2 for i in range(m)]
        nn = len(odList) - m
        for _ in range(nn):
            Na.append(norm.rvs(loc=1, scale=np.sqrt(new_sa2)))
            Nb.append(norm.rvs(loc=0, scale=np.sqrt(new_sb2)))
        logh = 0
        loghstar = 0
        logh += np.log(invgamma.pdf(new_sa2, 1))
        logh += np.log(invgamma.pdf(new_sb2, 1))
        loghstar += np.log(invgamma.pdf(old_sa2, 1))
        loghstar += np.log(invgamma.pdf(old_sb2, 1))
        for i in range(m, nn):                                             # :1084-1086 (non-log pdf, as coded)
            logh += norm.pdf(Na[i], loc=1, scale=np.sqrt(new_sa2))
            logh += norm.pdf(Nb[i], loc=0, scale=np.sqrt(new_sb2))
        for i in range(len(Ua)):
            logh += np.log(norm.pdf(Ua[i], loc=0, scale=np.sqrt(new_sa2)))
            logh += np.log(norm.pdf(Ub[i], loc=0, scale=np.sqrt(new_sb2)))
        for i in range(len(NUa)):
            loghstar += np.log(norm.pdf(NUa[i], loc=0, scale=np.sqrt(old_sa2)))
            loghstar += np.log(norm.pdf(NUb[i], loc=0, scale=np.sqrt(old_sb2)))
        hratio = np.exp(loghstar - logh)
        detjacob = 1 / np.power(2, 2 * m)
        for i in range(len(odList)):
            Tree[odList[i]].a = Na[i]
            Tree[odList[i]].b = Nb[i]
        return [hratio, detjacob, new_sa2, new_sb2]

    new_sa2 = invgamma.rvs(1)                                              # :1127-1128
    new_sb2 = invgamma.rvs(1)
    Na, Nb = [], []
    for _ in range(len(odList)):
        Na.append(norm.rvs(loc=1, scale=np.sqrt(new_sa2)))
        Nb.append(norm.rvs(loc=0, scale=np.sqrt(new_sb2)))
    for i in range(len(odList)):
        Tree[odList[i]].a = Na[i]
        Tree[odList[i]].b = Nb[i]
    return [new_sa2, new_sb2]


def newprop(Roots, count, sigma, y, indata, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b,
            faithful=False, trace=None):
    """One MH proposal (codes/funcs.py:1184-1306) -> [accepted, sigma, Root, sigma_a, sigma_b]."""
    K = len(Roots)
    Root = copy.deepcopy(Roots[count])
    info = {}
    [oldRoot, Root, lnPointers, change, Q, Qinv, last_a, last_b, cnode] = prop(
        Root, n_feature, Ops, Op_weights, Op_type, beta, sigma_a, sigma_b, info)
    sig = 4
    new_sigma = invgamma.rvs(sig)
    new_outputs = np.zeros((len(y), K))
    old_outputs = np.zeros((len(y), K))
    aux = auxprop(change, oldRoot, Root, lnPointers, sigma_a, sigma_b, last_a, last_b, cnode)
    if change in ('shrinkage', 'expansion'):
        hratio, detjacob, new_sa2, new_sb2 = aux
    else:
        new_sa2, new_sb2 = aux
        hratio = detjacob = None
    if trace is not None:
        trace.update(action=info["action"], change=change, Q=Q, Qinv=Qinv, new_sigma=new_sigma,
                     new_sa2=new_sa2, new_sb2=new_sb2, hratio=hratio, detjacob=detjacob,
                     proposed=Root, count=count)

    with np.errstate(all="ignore"):
        for i in range(K):
            if i == count:
                new_outputs[:, i] = allcal(Root, indata, faithful)[:, 0]
                old_outputs[:, i] = allcal(oldRoot, indata, faithful)[:, 0]
            else:
                col = allcal(Roots[i], indata, faithful)[:, 0]
                new_outputs[:, i] = col
                old_outputs[:, i] = col
        rank = np.linalg.matrix_rank(new_outputs)                          # :1226 (raises on NaN)
    if trace is not None:
        trace.update(rank=int(rank), new_maxabs=np.max(np.abs(new_outputs)))
    if rank < K:
        return [False, sigma, copy.deepcopy(oldRoot), sigma_a, sigma_b]

    G = (n_feature, Ops, Op_weights, Op_type, beta)
    with np.errstate(all="ignore"):
        yllstar = yloglike(y, new_outputs, new_sigma)
        yll = yloglike(y, old_outputs, sigma)
        log_yratio = yllstar - yll
        if change in ('shrinkage', 'expansion'):
            s_new = fstruc(Root, *G, new_sa2, new_sb2)
            s_old = fstruc(oldRoot, *G, sigma_a, sigma_b)
            log_struc = (s_old[0] + s_old[1]) - (s_new[0] + s_new[1])
            log_q = np.log(max(1e-5, Qinv / Q))
            logR = log_yratio + log_struc + log_q + np.log(max(1e-5, hratio)) + np.log(max(1e-5, detjacob))
        else:
            s_new = fstruc(Root, *G, new_sa2, new_sb2)[0]
            s_old = fstruc(oldRoot, *G, sigma_a, sigma_b)[0]
            log_struc = s_old - s_new
            log_q = np.log(max(1e-5, Qinv / Q))
            logR = log_yratio + log_struc + log_q
        logR = logR + np.log(invgamma.pdf(new_sigma, sig)) - np.log(invgamma.pdf(sigma, sig))
        alpha = min(logR, 0)
        u = np.random.uniform(low=0, high=1, size=1)[0]
        reject = np.log(u) >= alpha
    if trace is not None:
        trace.update(yllstar=yllstar, yll=yll, logR=logR, accept_u=u)
    if reject:
        return [False, sigma, copy.deepcopy(oldRoot), sigma_a, sigma_b]
    return [True, new_sigma, copy.deepcopy(Root), new_sa2, new_sb2]


# ------------------------------------------------------------------ chain loop (BSR.fit body)
def run_chain(X, y, K=3, beta=-1, val=100, faithful=False, max_props=None, on_proposal=None, ops=None, weights=None,
              arity=None):
    """One pass of the `while len(trainERRS) < MM` body (codes/bsr_class.py:99-273).

    Returns dict(roots, beta, errs, n_props, init_roots)."""
    if pd is not None and isinstance(X, np.ndarray):
        X = pd.DataFrame(X)
    n_feature = X.shape[1]
    n_train = X.shape[0]
    Ops, W, T = list(OPS), list(OP_WEIGHTS), list(OP_ARITY)                # codes/bsr_class.py:110-112 ...
    if ops is not None:                                                     # ... or the caller's table
        Ops = list(ops)
        W = list(weights) if weights is not None else [1.0 / len(Ops)] * len(Ops)
        known = dict(zip(OPS + EXT_OPS, OP_ARITY + EXT_ARITY))
        T = list(arity) if arity is not None else [known[o] for o in Ops]
    elif weights is not None:
        W = list(weights)
    RootLists = [[] for _ in range(K)]
    Siga, Sigb = [], []
    sigma = invgamma.rvs(1)
    for k in range(K):
        root = ONode(0)
        sa = invgamma.rvs(1)
        sb = invgamma.rvs(1)
        grow(root, n_feature, Ops, W, T, beta, sa, sb)
        RootLists[k].append(copy.deepcopy(root))
        Siga.append(sa)
        Sigb.append(sb)
    init_roots = [RootLists[k][-1] for k in range(K)]
    with np.errstate(all="ignore"):
        cols = np.zeros((n_train, K))
        for k in range(K):
            cols[:, k] = allcal(RootLists[k][-1], X, faithful)[:, 0]
        Beta, _ = intercept_fit(y, cols)
    total = 0
    n_props = 0
    errs = []
    Roots = []
    stop = False
    while total < val and not stop:
        for k in range(K):
            Roots = [RootLists[c][-1] for c in range(K)]
            tr = {} if on_proposal is not None else None
            res, sigma, Root, sa, sb = newprop(Roots, k, sigma, y, X, n_feature, Ops, W, T, beta,
                                               Siga[k], Sigb[k], faithful=faithful, trace=tr)
            n_props += 1
            total += 1
            Siga[k] = sa
            Sigb[k] = sb
            if res is True:
                RootLists[k].append(copy.deepcopy(Root))
                with np.errstate(all="ignore"):
                    cols = np.zeros((n_train, K))
                    for c in range(K):
                        cols[:, c] = allcal(RootLists[c][-1], X, faithful)[:, 0]
                    Beta, rmse = intercept_fit(y, cols)
                errs.append(rmse)
                total = 0
            if on_proposal is not None:
                tr.update(accepted=bool(res), sigma_out=sigma, sa_out=sa, sb_out=sb,
                          result=Root if res else None)
                on_proposal(tr)
            m = min(10, len(errs))
            if len(errs) > 100 and 1 - np.min(errs[-m:]) / np.mean(errs[-m:]) < 0.05:
                stop = True
                break
            if max_props is not None and n_props >= max_props:
                stop = True
                break
    return {"roots": Roots, "beta": Beta, "errs": errs, "n_props": n_props, "init_roots": init_roots}


def fit(X, y, K=3, itrNum=1, beta=-1, val=100, faithful=False):
    """BSR.fit restated (codes/bsr_class.py:77-278): itrNum independent chains, serial RNG stream."""
    roots, betas, errs, counts = [], [], [], []
    while len(errs) < itrNum:
        r = run_chain(X, y, K=K, beta=beta, val=val, faithful=faithful)
        roots.append(r["roots"])
        betas.append(r["beta"])
        errs.append(r["errs"])
        counts.append(r["n_props"])
    return {"roots_": roots, "betas_": betas, "train_err_": errs, "props_per_chain": counts}


def predict(roots, Beta, X, faithful=False):
    """BSR.predict (codes/bsr_class.py:53-68)."""
    if pd is not None and isinstance(X, np.ndarray):
        X = pd.DataFrame(X)
    n = X.shape[0]
    cols = np.zeros((n, len(roots)))
    with np.errstate(all="ignore"):
        for k, r in enumerate(roots):
            cols[:, k] = allcal(r, X, faithful)[:, 0]
    XX = np.concatenate((np.ones((n, 1)), cols), axis=1)
    return np.matmul(XX, Beta)


# ------------------------------------------------------------------ scoring only (the GPU checker)
def score_proposal(cur_cols, k, new_col, y, sigma_new):
    """Data-side result of one proposal, given the K current columns and the candidate column.

    Returns dict(rank, loglik, sse, scale, beta): the quantities newProp needs from the data
    (codes/funcs.py:1212-1235): rank gate on new_outputs, then ylogLike(y, new_outputs, new_sigma)."""
    M = np.array(cur_cols, dtype=np.float64, copy=True)
    M[:, k] = new_col
    out = {"maxabs": float(np.max(np.abs(M))) if np.all(np.isfinite(M)) else float("inf")}
    with np.errstate(all="ignore"):
        try:
            out["rank"] = int(np.linalg.matrix_rank(M))
        except np.linalg.LinAlgError:
            out["rank"] = -1
            return out
        if out["rank"] == M.shape[1]:
            ll, sse, scale, beta = yloglike_parts(np.asarray(y), M, sigma_new)
            out.update(loglik=float(ll), sse=float(sse), scale=float(scale), beta=beta[:, 0].copy())
    return out
