"""Host-side proposal generator of the BSR sampler: grow / fStruc / Prop / auxProp.

This is the scalar driver that decides WHAT the GPU scores (SURVEY.md 8 row a-H).  It is not data-parallel and stays
on the host, but the "accepted-tree sequence bit-exact for a fixed seed" criterion pins its decisions and its RNG
draw order, so every branch below follows the reference's behaviour, quirks included
(codes/funcs.py:74-119 grow, :349-398 fStruc, :406-923 Prop, :935-1138 auxProp).  Draws go through bsr.rng.
"""
import math

import numpy as np

from . import rng
from .node import Node, clone, genList, getNum, numLT, upDepth

LN = 'ln'


class OpTable:
    """Operator table of codes/bsr_class.py:110-112 with a pre-built sampler for np.random.choice(p=weights)."""

    _cache = {}

    def __init__(self, Ops, Op_weights, Op_type):
        self.ops = list(Ops)
        self.weights = list(Op_weights)
        self.types = list(Op_type)
        self.choose = rng.Chooser(self.weights)
        self.logw = [rng.flog(w) for w in self.weights]

    @classmethod
    def get(cls, Ops, Op_weights, Op_type):
        key = (tuple(Ops), tuple(Op_weights), tuple(Op_type))
        t = cls._cache.get(key)
        if t is None:
            t = cls._cache[key] = cls(Ops, Op_weights, Op_type)
        return t


def default_table():
    from .node import OPS, OP_TYPE
    return OpTable.get(OPS, [1.0 / len(OPS)] * len(OPS), OP_TYPE)


# ------------------------------------------------------------------------------------------------ grow / fStruc
def grow_t(node, nfeature, T, beta, sigma_a, sigma_b):
    """Random growth from `node` (codes/funcs.py:74-119); draw order: uniform, [randint | choice], [2 normals]."""
    depth = node.depth
    pick = True
    if depth > 0:
        prob = 1 / math.pow(1 + depth, -beta)
        if rng.uniform() > prob:
            node.feature = rng.randint_arr(0, nfeature)   # :83, overwritten by the second draw at :99
            node.type = 0
            pick = False
    if pick:
        k = T.choose()
        node.operator = T.ops[k]
        node.type = T.types[k]
        node.op_ind = k
    if node.type == 0:
        node.feature = rng.randint_arr(0, nfeature)
    elif node.type == 1:
        node.left = Node(depth + 1)
        node.left.parent = node
        if node.operator == LN:
            node.a = rng.normal(1, math.sqrt(sigma_a))
            node.b = rng.normal(0, math.sqrt(sigma_b))
        grow_t(node.left, nfeature, T, beta, sigma_a, sigma_b)
    else:
        node.left = Node(depth + 1)
        node.left.parent = node
        node.right = Node(depth + 1)
        node.right.parent = node
        grow_t(node.left, nfeature, T, beta, sigma_a, sigma_b)
        grow_t(node.right, nfeature, T, beta, sigma_a, sigma_b)


def fstruc_t(node, n_feature, T, beta, sigma_a, sigma_b):
    """(log prior of the structure, log prior of the ln parameters) of the subtree (codes/funcs.py:349-398).
    Uses each node's STORED depth and op_ind, as the reference does."""
    ls = 0
    lp = 0
    if node.type == 0:
        ls += rng.flog(1 - 1 / math.pow(1 + node.depth, -beta))
        ls -= math.log(n_feature)
    else:
        if node.depth == 0:
            ls += T.logw[node.op_ind]
        else:
            ls += math.log(1 + node.depth) * beta + T.logw[node.op_ind]
        if node.type == 1 and node.operator == LN:
            lp -= math.pow(node.a - 1, 2) / (2 * sigma_a)
            lp -= math.pow(node.b, 2) / (2 * sigma_b)
            lp -= 0.5 * math.log(2 * math.pi * sigma_a)
            lp -= 0.5 * math.log(2 * math.pi * sigma_b)
    if node.left is None:
        return ls, lp
    a, b = fstruc_t(node.left, n_feature, T, beta, sigma_a, sigma_b)
    ls += a
    lp += b
    if node.right is None:
        return ls, lp
    a, b = fstruc_t(node.right, n_feature, T, beta, sigma_a, sigma_b)
    ls += a
    lp += b
    return ls, lp


# ------------------------------------------------------------------------------------------------ Prop
def _detr_candidates(tree):
    """Non-terminal nodes, except a root whose children are all terminal (codes/funcs.py:454-468)."""
    out = []
    for n in tree:
        if n.type == 0:
            continue
        if n.parent is None:
            if n.right is None:
                if n.left.type == 0:
                    continue
            elif n.left.type == 0 and n.right.type == 0:
                continue
        out.append(n)
    return out


def _swap_child(parent, old, new):
    if parent.left is old:
        parent.left = new
    else:
        parent.right = new
    new.parent = parent


class Move:
    """Outcome of one structural proposal (the 9-list of codes/funcs.py:923 plus the action name)."""

    __slots__ = ("root", "ln_nodes", "change", "Q", "Qinv", "last_a", "last_b", "cnode", "action")


def prop_inplace(Root, n_feature, T, beta, sigma_a, sigma_b):
    """Edits the tree rooted at `Root` in place (the caller passes a private copy) -> Move.

    Follows codes/funcs.py:406-923 move by move; comments give the reference lines."""
    G = (n_feature, T, beta, sigma_a, sigma_b)
    tree = genList(Root)
    ln_nodes = [n for n in tree if n.operator == LN]
    mv = Move()
    mv.ln_nodes = ln_nodes
    mv.last_a = [n.a for n in ln_nodes]
    mv.last_b = [n.b for n in ln_nodes]
    mv.cnode = None
    term = [n for n in tree if n.type == 0]
    nterm = [n for n in tree if n.type != 0]
    ltNum = len(ln_nodes)
    change = ''
    Q = Qinv = 1
    detcd = _detr_candidates(tree)

    p_stay = 0.25 * ltNum / (ltNum + 3)                                       # :475-480
    p_grow = (1 - p_stay) * min(1, 4 / (len(nterm) + 2)) / 3
    p_prune = (1 - p_stay) / 3 - p_grow
    p_detr = (1 - p_stay) * (1 / 3) * len(detcd) / (3 + len(detcd))
    p_trans = (1 - p_stay) / 3 - p_detr
    p_rop = (1 - p_stay) / 6

    u = rng.uniform()                                                          # :483

    if u <= p_stay:                                                            # :490-500
        action = 'stay'
        Q = Qinv = p_stay
        sa, sb = math.sqrt(sigma_a), math.sqrt(sigma_b)
        for n in ln_nodes:
            n.a = rng.normal(1, sa)
            n.b = rng.normal(1, sb)

    elif u <= p_stay + p_grow:                                                 # :503-536
        action = 'grow'
        tgt = term[rng.randint(0, len(term))]
        grow_t(tgt, *G)
        if tgt.type != 0:
            fs = fstruc_t(tgt, *G)[0]
            Q = p_grow * rng.fexp(fs) / len(term)
            new_lt = numLT(Root)
            new_n = getNum(Root)
            nt = sum(1 for n in genList(Root) if n.type == 0)
            new_p = (1 - 0.25 * new_lt / (new_lt + 3)) * (1 - min(1, 4 / ((new_n - nt) + 2))) / 3
            Qinv = new_p / max(1, (new_n - nt - 1))
            if new_lt > ltNum:
                change = 'expansion'

    elif u <= p_stay + p_grow + p_prune:                                       # :539-579
        action = 'prune'
        tgt = nterm[rng.randint(1, len(nterm))]
        fs = fstruc_t(tgt, *G)[0]
        if numLT(tgt) > 0:
            change = 'shrinkage'
        tgt.left = None
        tgt.right = None
        tgt.operator = None
        tgt.type = 0
        tgt.feature = rng.randint_arr(0, n_feature)
        new_lt = numLT(Root)
        new_tree = genList(Root)
        nt = sum(1 for n in new_tree if n.type == 0)
        Q = p_prune / ((len(nterm) - 1) * n_feature)
        pg = 1 - 0.25 * new_lt / (new_lt + 3) * 0.75 * min(1, 4 / ((len(new_tree) - nt) + 2))
        Qinv = pg * rng.fexp(fs) / nt

    elif u <= p_stay + p_grow + p_prune + p_detr:                              # :582-673
        action = 'detransform'
        dn = detcd[rng.randint(0, len(detcd))]
        cut = None
        Q = p_detr / len(detcd)
        if dn.parent is None:
            if dn.right is None:
                Root = Root.left
            elif dn.left.type == 0:
                cut = Root.left
                Root = Root.right
            elif dn.right.type == 0:
                cut = Root.right
                Root = Root.left
            else:
                if rng.uniform() <= 0.5:
                    cut = Root.right
                    Root = Root.left
                else:
                    cut = Root.left
                    Root = Root.right
                Q = Q / 2
        elif dn.type == 1:
            _swap_child(dn.parent, dn, dn.left)
        else:
            if rng.uniform() <= 0.5:
                cut = dn.right
                _swap_child(dn.parent, dn, dn.left)
            else:
                cut = dn.left
                _swap_child(dn.parent, dn, dn.right)
            Q = Q / 2
        Root.parent = None
        upDepth(Root)
        new_tree = genList(Root)
        new_lt = sum(1 for n in new_tree if n.operator == LN)
        if new_lt < ltNum:
            change = 'shrinkage'
        new_pstay = 0.25 * new_lt / (new_lt + 3)
        nd = len(_detr_candidates(new_tree))
        new_pdetr = (1 - new_pstay) * (1 / 3) * nd / (nd + 3)
        new_ptr = (1 - new_pstay) / 3 - new_pdetr
        Qinv = new_ptr * T.weights[dn.op_ind] / len(new_tree)
        if cut is not None:
            Qinv = Qinv * rng.fexp(fstruc_t(cut, *G)[0])                       # cut keeps its stale depths

    elif u <= p_stay + p_grow + p_prune + p_detr + p_trans:                    # :679-786
        action = 'transform'
        ins = tree[rng.randint(0, len(tree))]
        k = T.choose()
        w = T.weights[k]
        nn = Node(ins.depth)
        nn.operator = T.ops[k]
        nn.type = T.types[k]
        nn.op_ind = k
        if nn.type == 1 and nn.operator == LN:
            change = 'expansion'
        par = ins.parent
        if par is None:
            Root = nn
        else:
            if par.left is ins:
                par.left = nn
            else:
                par.right = nn
            nn.parent = par
        nn.left = ins
        ins.parent = nn
        if nn.type == 1:
            upDepth(Root)
            Q = p_trans * w / len(tree)
        else:
            nr = Node(nn.depth + 1)
            nn.right = nr
            nr.parent = nn
            upDepth(Root)
            grow_t(nr, *G)
            Q = p_trans * w * rng.fexp(fstruc_t(nr, *G)[0]) / len(tree)
        new_tree = genList(Root)
        new_lt = sum(1 for n in new_tree if n.operator == LN)
        if new_lt > ltNum:
            change = 'expansion'
        new_pstay = 0.25 * new_lt / (new_lt + 3)
        nd = len(_detr_candidates(new_tree))
        new_pdetr = (1 - new_pstay) * (1 / 3) * nd / (nd + 3)
        Qinv = new_pdetr / nd
        if nn.type == 2 and nn.left.type > 0 and nn.right.type > 0:
            Qinv = Qinv / 2

    elif u <= p_stay + p_grow + p_prune + p_detr + p_trans + p_rop:            # :791-903
        action = 'ReassignOperator'
        cn = nterm[rng.randint(0, len(nterm))]
        mv.cnode = cn
        last_op, last_oi, last_type = cn.operator, cn.op_ind, cn.type
        k = T.choose()
        new_op, new_type = T.ops[k], T.types[k]
        if last_type == 1 and new_type == 1:                                   # unary -> unary (op_ind not updated)
            cn.operator = new_op
            if last_op == LN:
                if new_op != LN:
                    cn.a = None
                    cn.b = None
                    change = 'shrinkage'
            elif new_op == LN:
                change = 'expansion'
            Q = T.weights[k]
            Qinv = T.weights[last_oi]
        elif last_type == 1:                                                   # unary -> binary
            cn.operator = new_op
            cn.type = 2
            if last_op == LN:
                cn.a = None
                cn.b = None
            cn.right = Node(cn.depth + 1)
            cn.right.parent = cn
            grow_t(cn.right, *G)
            fs = fstruc_t(cn.right, *G)[0]
            Q = p_rop * rng.fexp(fs) * T.weights[k] / len(nterm)
            new_n = getNum(Root)
            nt = sum(1 for n in genList(Root) if n.type == 0)
            new_lt = numLT(Root)
            new_p0 = new_lt / (4 * (new_lt + 3))
            Qinv = 0.125 * (1 - new_p0) * T.weights[last_oi] / (new_n - nt)
            if new_lt > ltNum:
                change = 'expansion'
            elif new_lt < ltNum:
                change = 'shrinkage'
        elif new_type == 1:                                                    # binary -> unary
            cut = cn.right                                                     # reference deep-copies; detaching is enough
            p_lt = numLT(cut)
            if p_lt > 1:
                change = 'shrinkage'
            elif new_op == LN and p_lt == 0:
                change = 'expansion'
            cn.right = None
            cn.operator = new_op
            cn.type = new_type
            Q = p_rop * T.weights[k] / len(nterm)
            new_n = getNum(Root)
            genList(Root)
            new_lt = numLT(Root)
            new_p0 = new_lt / (4 * (new_lt + 3))
            fs = fstruc_t(cut, *G)[0]
            Qinv = 0.125 * (1 - new_p0) * rng.fexp(fs) * T.weights[last_oi] / new_n   # newTerm empty at :893-894
        else:                                                                  # binary -> binary
            cn.operator = new_op
            Q = T.weights[k]
            Qinv = T.weights[last_oi]

    else:                                                                      # :907-917
        action = 'ReassignFeature'
        tgt = term[rng.randint(0, len(term))]
        tgt.feature = rng.randint_arr(0, n_feature)
        Q = Qinv = 1

    Root.parent = None
    upDepth(Root)
    mv.root = Root
    mv.change = change
    mv.Q = Q
    mv.Qinv = Qinv
    mv.action = action
    return mv


# ------------------------------------------------------------------------------------------------ auxProp
def aux_inplace(change, Root, ln_nodes, sigma_a, sigma_b, last_a, last_b):
    """Samples sigma_a2/sigma_b2 and rewrites every ln node's (a,b) (codes/funcs.py:935-1138).
    Returns (new_sa2, new_sb2, hratio, detjacob); the last two are None when the ln count did not change."""
    tree = genList(Root)
    lns = [n for n in tree if n.operator == LN]
    new_sa2 = rng.invgamma_rvs(1)                                              # :945-946
    new_sb2 = rng.invgamma_rvs(1)
    log = rng.flog

    if change == 'shrinkage':                                                  # :950-1026
        keep_a, keep_b, cut_a, cut_b = [], [], [], []
        for i, p in enumerate(ln_nodes):
            if p.operator == LN:           # includes nodes detached with a cut subtree: nobody reset them
                keep_a.append(last_a[i])
                keep_b.append(last_b[i])
            else:
                cut_a.append(last_a[i])
                cut_b.append(last_b[i])
        for i in range(len(lns) - len(keep_a)):
            keep_a.append(cut_a[i])
            keep_b.append(cut_b[i])
        n0 = len(keep_a)
        sa, sb = math.sqrt(new_sa2), math.sqrt(new_sb2)
        Ua, Ub = [], []
        for _ in range(n0):
            Ua.append(rng.normal(0, sa))
            Ub.append(rng.normal(0, sb))
        Na = [keep_a[i] + Ua[i] for i in range(n0)]
        Nb = [keep_b[i] + Ub[i] for i in range(n0)]
        NUa = [keep_a[i] - Ua[i] for i in range(n0)] + list(last_a)
        NUb = [keep_b[i] - Ub[i] for i in range(n0)] + list(last_b)
        logh = 0
        loghstar = 0
        logh += log(rng.invgamma_pdf(new_sa2, 1))
        logh += log(rng.invgamma_pdf(new_sb2, 1))
        loghstar += log(rng.invgamma_pdf(sigma_a, 1))
        loghstar += log(rng.invgamma_pdf(sigma_b, 1))
        for i in range(n0):
            logh += log(rng.norm_pdf(Ua[i], 0, sa))
            logh += log(rng.norm_pdf(Ub[i], 0, sb))
        osa, osb = math.sqrt(sigma_a), math.sqrt(sigma_b)
        for i in range(len(NUa)):
            loghstar += log(rng.norm_pdf(NUa[i], 0, osa))
            loghstar += log(rng.norm_pdf(NUb[i], 0, osb))
        hratio = rng.fexp(loghstar - logh)
        detjacob = 2 ** (2 * n0)
        for i, n in enumerate(lns):
            n.a = Na[i]
            n.b = Nb[i]
        return new_sa2, new_sb2, hratio, detjacob

    if change == 'expansion':                                                  # :1030-1110
        new_sa2 = rng.invgamma_rvs(1)
        new_sb2 = rng.invgamma_rvs(1)
        m = len(last_a)
        sa, sb = math.sqrt(new_sa2), math.sqrt(new_sb2)
        Ua, Ub = [], []
        for _ in range(m):
            Ua.append(rng.normal(0, sa))
            Ub.append(rng.normal(0, sb))
        Na = [(last_a[i] + Ua[i]) / 2 for i in range(m)]
        Nb = [(last_b[i] + Ub[i]) / 2 for i in range(m)]
        NUa = [(last_a[i] - Ua[i]) / 2 for i in range(m)]
        NUb = [(last_b[i] - Ub[i]) / 2 for i in range(m)]
        nn = len(lns) - m
        for _ in range(nn):
            Na.append(rng.normal(1, sa))
            Nb.append(rng.normal(0, sb))
        logh = 0
        loghstar = 0
        logh += log(rng.invgamma_pdf(new_sa2, 1))
        logh += log(rng.invgamma_pdf(new_sb2, 1))
        loghstar += log(rng.invgamma_pdf(sigma_a, 1))
        loghstar += log(rng.invgamma_pdf(sigma_b, 1))
        for i in range(m, nn):                                                 # :1084-1086 adds plain pdf values
            logh += rng.norm_pdf(Na[i], 1, sa)
            logh += rng.norm_pdf(Nb[i], 0, sb)
        for i in range(m):
            logh += log(rng.norm_pdf(Ua[i], 0, sa))
            logh += log(rng.norm_pdf(Ub[i], 0, sb))
        osa, osb = math.sqrt(sigma_a), math.sqrt(sigma_b)
        for i in range(m):
            loghstar += log(rng.norm_pdf(NUa[i], 0, osa))
            loghstar += log(rng.norm_pdf(NUb[i], 0, osb))
        hratio = rng.fexp(loghstar - logh)
        detjacob = 1 / (2 ** (2 * m))
        for i, n in enumerate(lns):
            n.a = Na[i]
            n.b = Nb[i]
        return new_sa2, new_sb2, hratio, detjacob

    new_sa2 = rng.invgamma_rvs(1)                                              # :1127-1128
    new_sb2 = rng.invgamma_rvs(1)
    sa, sb = math.sqrt(new_sa2), math.sqrt(new_sb2)
    vals = []
    for _ in lns:
        a = rng.normal(1, sa)
        b = rng.normal(0, sb)
        vals.append((a, b))
    for n, (a, b) in zip(lns, vals):
        n.a = a
        n.b = b
    return new_sa2, new_sb2, None, None


# ------------------------------------------------------------------------------------------------ log ratio
SIG_SHAPE = 4   # codes/funcs.py:1194


def _pymax(a, b):
    """Python's max(a, b) including its NaN behaviour (first argument wins unless b > a)."""
    return b if b > a else a


def log_ratio(change, Q, Qinv, hratio, detjacob, yllstar, yll, s_new, s_old, new_sigma, sigma):
    """log R of codes/funcs.py:1230-1296.  s_new / s_old are the (structure, parameter) log priors of the proposed
    and current trees; the no-jump branch uses the structure part only (:1287-1289)."""
    log = rng.flog
    log_y = yllstar - yll
    log_q = log(_pymax(1e-5, rng.fdiv(Qinv, Q)))
    if change in ('shrinkage', 'expansion'):
        log_s = (s_old[0] + s_old[1]) - (s_new[0] + s_new[1])
        logR = log_y + log_s + log_q + log(_pymax(1e-5, hratio)) + log(_pymax(1e-5, detjacob))
    else:
        log_s = s_old[0] - s_new[0]
        logR = log_y + log_s + log_q
    logR = logR + log(rng.invgamma_pdf(new_sigma, SIG_SHAPE)) - log(rng.invgamma_pdf(sigma, SIG_SHAPE))
    return logR


def accept_test(logR, u):
    """`np.log(test) >= alpha` means reject (codes/funcs.py:1298-1306); NaN logR accepts, as in the reference."""
    alpha = 0 if 0 < logR else logR   # Python's min(logR, 0)
    return not (rng.flog(u) >= alpha)


# ------------------------------------------------------------------------------------------------ value ranges
def tree_range(node, xlo, xhi, N):
    """Interval bound (lo, hi) of a tree's values over the per-feature ranges of X: what max|z| can be.  Used by the
    chain engines to guess the rank gate's verdict on a candidate whose scale is far from its siblings'
    (bsr.chain.Chain._predict_reject; the C++ sampler has the same function)."""
    inf = float("inf")
    if node.type == 0:
        f = int(np.asarray(node.feature).reshape(-1)[0])
        return float(xlo[f]), float(xhi[f])
    lo, hi = tree_range(node.left, xlo, xhi, N)

    def inv(l, h):
        if l <= 0 <= h:
            m = 2 * N / max(h - l, 1e-300)
            return -m, m
        return min(1 / l, 1 / h), max(1 / l, 1 / h)

    def mul(a, b):
        c = [a[0] * b[0], a[0] * b[1], a[1] * b[0], a[1] * b[1]]
        c = [0.0 if v != v else v for v in c]
        return min(c), max(c)

    def cube(v):
        try:
            return v ** 3
        except OverflowError:
            return math.copysign(inf, v)
    op = node.operator
    if node.type == 1:
        if op == LN:
            p, q = node.a * lo + node.b, node.a * hi + node.b
            return min(p, q), max(p, q)
        if op == 'neg':
            return -hi, -lo
        if op in ('sin', 'cos'):
            return -1.0, 1.0
        if op == 'exp':
            def ex(v):
                if v > 200:
                    return 1e10
                return math.exp(v)
            return ex(lo), max(ex(hi), ex(min(hi, 200.0)))
        if op == 'square':
            try:
                a, b = lo * lo, hi * hi
            except OverflowError:
                return 0.0, inf
            return (0.0 if lo <= 0 <= hi else min(a, b)), max(a, b)
        if op == 'cubic':
            return cube(lo), cube(hi)
        if op == 'inv':
            return inv(lo, hi)
        if op == 'log':
            top = math.log(max(abs(lo), abs(hi), 1e-300))
            bot = math.log(max((hi - lo) / (2 * N), 1e-300)) if lo <= 0 <= hi else math.log(min(abs(lo), abs(hi)))
            return min(bot, top), top
        return lo, hi
    rl, rh = tree_range(node.right, xlo, xhi, N)
    if op == '+':
        return lo + rl, hi + rh
    if op == 'sub':
        return lo - rh, hi - rl
    if op == 'div':
        return mul((lo, hi), inv(rl, rh))
    return mul((lo, hi), (rl, rh))


def canon_key(node, top=True):
    """Structural key of the column a tree computes: equal keys = equal columns up to sign (children of + and * order-free,
    a negation at the root dropped).  The chain engines guess the rank gate with it: a candidate that repeats a sibling,
    siblings that repeat each other."""
    if top:
        while node.type == 1 and node.operator == 'neg':
            node = node.left
    if node.type == 0:
        return ("x", int(np.asarray(node.feature).reshape(-1)[0]))
    if node.type == 1:
        par = (float(node.a), float(node.b)) if node.operator == LN else ()
        return (node.operator, par, canon_key(node.left, False))
    l, r = canon_key(node.left, False), canon_key(node.right, False)
    if node.operator in ('+', '*'):
        l, r = sorted((l, r), key=repr)
    return (node.operator, l, r)
