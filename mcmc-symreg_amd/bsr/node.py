"""Expression-tree node type and structural helpers of the BSR sampler.

Keeps the reference's `Node` surface (codes/funcs.py:30-67) so that trees built by user code, `roots_`
and the helper functions re-exported by the reference package (codes/__init__.py:9-11) keep working:
`genList`, `getHeight`, `getNum`, `numLT`, `upDepth`, `Express`, `display`, `shrink`, `upgOd`.
Nodes never carry an (N,1) data array here: columns live on the GPU.
"""
import numpy as np

OPS = ['inv', 'ln', 'neg', 'sin', 'cos', 'exp', 'square', 'cubic', '+', '*']   # codes/bsr_class.py:110
OP_TYPE = [1, 1, 1, 1, 1, 1, 1, 1, 2, 2]                                        # codes/bsr_class.py:112
OP_CODE = {name: i for i, name in enumerate(OPS)}
# Extensions beyond the reference's table (opcodes of include/bsr_hip.h, semantics in the oracle):
# sub x-y, div where(y==0, 0, x/y), log where(x==0, 0, log|x|).  Usable through BSR(ops=..., op_weights=...).
EXT_OPS = ['sub', 'div', 'log']
OP_CODE.update({'sub': 13, 'div': 14, 'log': 15})
OP_ARITY = {name: OP_TYPE[i] for i, name in enumerate(OPS)}
OP_ARITY.update({'sub': 2, 'div': 2, 'log': 1})
OP_NAME = {code: name for name, code in OP_CODE.items()}
COMMUTATIVE = ('+', '*')


class Operator:
    """name / function / arity holder (codes/funcs.py:23-27; unused by the sampler)."""

    def __init__(self, name, function, arity):
        self.name = name
        self.func = function
        self.arity = arity


class Node:
    """type: -1 not grown, 0 terminal, 1 unary, 2 binary (codes/funcs.py:30-55)."""

    __slots__ = ("type", "order", "left", "right", "depth", "parent", "operator", "op_ind", "data", "feature",
                 "a", "b")

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

    def inform(self):  # codes/funcs.py:56-67
        print("order:", self.order)
        print("type:", self.type)
        print("depth:", self.depth)
        print("operator:", self.operator)
        print("data:", self.data)
        print("feature:", self.feature)
        if self.operator == 'ln':
            print(" ln_a:", self.a)
            print(" ln_b:", self.b)


def clone(node, parent=None):
    """Structural deep copy (what copy.deepcopy gives the reference, minus the data arrays)."""
    top = Node(node.depth)
    work = [(node, top, parent)]
    while work:
        src, dst, par = work.pop()
        dst.type = src.type
        dst.order = src.order
        dst.depth = src.depth
        dst.parent = par
        dst.operator = src.operator
        dst.op_ind = src.op_ind
        dst.feature = src.feature
        dst.a = src.a
        dst.b = src.b
        if src.left is not None:
            dst.left = Node(src.left.depth)
            work.append((src.left, dst.left, dst))
        if src.right is not None:
            dst.right = Node(src.right.depth)
            work.append((src.right, dst.right, dst))
    return top


def genList(node):
    """Pre-order list (node, left subtree, right subtree); refreshes .order (codes/funcs.py:127-142)."""
    out = []
    work = [node]
    while work:
        n = work.pop()
        out.append(n)
        if n.left is not None:
            if n.right is not None:
                work.append(n.right)
            work.append(n.left)
    for i, n in enumerate(out):
        n.order = i
    return out


def upgOd(Tree):  # codes/funcs.py:166-169
    for i, n in enumerate(Tree):
        n.order = i


def shrink(node):  # codes/funcs.py:149-159
    if node.left is None:
        print("Already a terminal node!")
    else:
        node.left = None
        node.right = None
        node.type = 0
        node.operator = None
        node.a = None
        node.b = None


def getHeight(node):  # codes/funcs.py:255-263
    best = 0
    work = [(node, 0)]
    while work:
        n, h = work.pop()
        if n.type == 0:
            best = max(best, h)
        elif n.type == 1:
            work.append((n.left, h + 1))
        else:
            work.append((n.left, h + 1))
            work.append((n.right, h + 1))
    return best


def getNum(node):  # codes/funcs.py:269-277
    cnt = 0
    work = [node]
    while work:
        n = work.pop()
        cnt += 1
        if n.type == 1:
            work.append(n.left)
        elif n.type != 0:
            work.append(n.left)
            work.append(n.right)
    return cnt


def numLT(node):  # codes/funcs.py:283-292
    cnt = 0
    work = [node]
    while work:
        n = work.pop()
        if n.type == 1:
            if n.operator == 'ln':
                cnt += 1
            work.append(n.left)
        elif n.type != 0:
            work.append(n.left)
            work.append(n.right)
    return cnt


def upDepth(Root):  # codes/funcs.py:298-307
    Root.depth = 0 if Root.parent is None else Root.parent.depth + 1
    work = [Root]
    while work:
        n = work.pop()
        if n.left is not None:
            n.left.depth = n.depth + 1
            work.append(n.left)
            if n.right is not None:
                n.right.depth = n.depth + 1
                work.append(n.right)


def Express(node):
    """String form used by BSR.model() (codes/funcs.py:314-342)."""
    if node.type == 0:
        return "x" + str(node.feature)
    if node.type == 1:
        inner = Express(node.left)
        op = node.operator
        if op == 'exp':
            return "exp(" + inner + ")"
        if op == 'ln':
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
        return Express(node.left) + "+" + Express(node.right)
    if node.operator == 'sub':
        return "(" + Express(node.left) + ")-(" + Express(node.right) + ")"
    if node.operator == 'div':
        return "(" + Express(node.left) + ")/[" + Express(node.right) + "]"
    return "(" + Express(node.left) + ")*(" + Express(node.right) + ")"


def display(Tree):
    """Prints operators / features level by level (codes/funcs.py:227-246)."""
    levels = {}
    for n in Tree:
        levels.setdefault(n.depth, []).append(n)
    for d in range(0, (max(levels) if levels else -1) + 1):
        st = " "
        for n in levels.get(d, []):
            st = st + (n.operator if n.type > 0 else str(n.feature)) + " "
        print(st)


def feature_index(node):
    return int(np.asarray(node.feature).reshape(-1)[0])
