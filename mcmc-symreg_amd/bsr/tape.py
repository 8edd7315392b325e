"""Flattening of Node trees into postfix (RPN) tapes of bsr_node rows (include/bsr_hip.h).

Row = (opcode, left, right, feature, a, b), 32 bytes.  Rows are in evaluation order for the GPU stack
machine; `left`/`right` are the row indices of the children, so the tree is recoverable from the tape.
For the commutative binary operators the child needing the deeper stack is emitted first (Sethi-Ullman),
which keeps the interpreter's stack within its register slots; x+y and x*y are bitwise commutative.
"""
import numpy as np

from .node import Node, OPS, OP_CODE, OP_TYPE, OP_ARITY, OP_NAME, COMMUTATIVE

NODE_DTYPE = np.dtype([("opcode", "<i4"), ("left", "<i4"), ("right", "<i4"), ("feature", "<i4"),
                       ("a", "<f8"), ("b", "<f8")], align=True)
assert NODE_DTYPE.itemsize == 32
OP_TERMINAL = 10
MAX_TAPE = 16384
MAX_STACK = 24


def flatten(root):
    """Node tree -> structured array of bsr_node rows in postfix order."""
    # pass 1: post-order list + stack need per node
    order = []
    work = [(root, False)]
    need = {}
    while work:
        n, seen = work.pop()
        if n.type == 0:
            need[id(n)] = 1
            continue
        if not seen:
            work.append((n, True))
            work.append((n.left, False))
            if n.type == 2:
                work.append((n.right, False))
            elif n.type != 1:
                raise ValueError("cannot flatten a node of type %r (tree not grown)" % (n.type,))
        else:
            if n.type == 1:
                need[id(n)] = need[id(n.left)]
            else:
                a, b = need[id(n.left)], need[id(n.right)]
                if n.operator in COMMUTATIVE:
                    need[id(n)] = a + 1 if a == b else max(a, b)
                else:                                   # left is evaluated first and waits on the stack
                    need[id(n)] = max(a, b + 1)
    # pass 2: emit
    rows = []
    index = {}
    work = [(root, False)]
    while work:
        n, seen = work.pop()
        if n.type == 0:
            index[id(n)] = len(rows)
            rows.append((OP_TERMINAL, -1, -1, int(np.asarray(n.feature).reshape(-1)[0]), 0.0, 0.0))
        elif not seen:
            work.append((n, True))
            if n.type == 1:
                work.append((n.left, False))
            else:
                first, second = (n.left, n.right)
                if n.operator in COMMUTATIVE and need[id(n.right)] > need[id(n.left)]:
                    first, second = n.right, n.left
                work.append((second, False))
                work.append((first, False))
        else:
            index[id(n)] = len(rows)
            code = OP_CODE[n.operator]
            if n.type == 1:
                a = float(n.a) if code == 1 else 0.0
                b = float(n.b) if code == 1 else 0.0
                rows.append((code, index[id(n.left)], -1, -1, a, b))
            else:
                rows.append((code, index[id(n.left)], index[id(n.right)], -1, 0.0, 0.0))
    return np.array(rows, dtype=NODE_DTYPE)


def unflatten(tape):
    """bsr_node rows -> Node tree (op_ind set to the opcode; depths refreshed)."""
    nodes = []
    for r in tape:
        op = int(r["opcode"])
        n = Node(0)
        if op == OP_TERMINAL:
            n.type = 0
            n.feature = np.array([int(r["feature"])])
        else:
            n.operator = OP_NAME[op]
            n.type = OP_ARITY[n.operator]
            n.op_ind = op if op < len(OPS) else None
            n.left = nodes[int(r["left"])]
            n.left.parent = n
            if n.type == 2:
                n.right = nodes[int(r["right"])]
                n.right.parent = n
            if op == 1:
                n.a = float(r["a"])
                n.b = float(r["b"])
        nodes.append(n)
    root = nodes[-1]
    root.parent = None
    from .node import upDepth
    upDepth(root)
    return root


def pack(tapes):
    """List of tapes -> (rows, offsets) as the C ABI wants them."""
    off = np.zeros(len(tapes) + 1, dtype=np.int32)
    for i, t in enumerate(tapes):
        off[i + 1] = off[i] + len(t)
    rows = np.concatenate(tapes) if tapes else np.zeros(0, dtype=NODE_DTYPE)
    return np.ascontiguousarray(rows), off


def signature(tape):
    """Opcode/feature sequence of a tape: the part of a tree that must match the reference bit-exactly."""
    return tuple(int(r["opcode"]) if int(r["opcode"]) != OP_TERMINAL else 100 + int(r["feature"]) for r in tape)
