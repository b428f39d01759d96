"""Constraint circuits with the reference's exact node-identity semantics.

The AIR evaluator of the reference is *generated at build time* from circuits
(/root/reference/triton-vm/build.rs:12-25); neither the generated code nor a Rust toolchain exists in
this container.  To evaluate the same polynomials in the same order on the GPU, the multicircuit has
to be rebuilt exactly: degree lowering is deterministic but depends on node ids (creation order),
on de-duplication and on constant folding (/root/reference/triton-constraint-circuit/src/lib.rs).

This module restates those semantics; function docstrings cite the lines they follow.
All field values are canonical python ints (not Montgomery words).
"""
import sys

P = 2**64 - 2**32 + 1
sys.setrecursionlimit(100000)

ADD, MUL = "+", "*"


def xfe_mul(a, b):
    """F_p[X]/(X^3 - X + 1), specification/src/isa.md:8"""
    d0 = a[0] * b[0]
    d1 = a[0] * b[1] + a[1] * b[0]
    d2 = a[0] * b[2] + a[1] * b[1] + a[2] * b[0]
    d3 = a[1] * b[2] + a[2] * b[1]
    d4 = a[2] * b[2]
    return ((d0 - d3) % P, (d1 + d3 - d4) % P, (d2 + d4) % P)


class Node:
    """ConstraintCircuit (lib.rs:364-369) with its CircuitExpression (lib.rs:293-299) inlined.
    kind: 'b' (BConst, val = int), 'x' (XConst, val = 3-tuple), 'in' (Input, val = indicator tuple),
    'ch' (Challenge, val = index), 'op' (BinOp, op/lhs/rhs)."""
    __slots__ = ("id", "kind", "val", "op", "lhs", "rhs")

    def __init__(self, id_, kind, val=None, op=None, lhs=None, rhs=None):
        self.id, self.kind, self.val, self.op, self.lhs, self.rhs = id_, kind, val, op, lhs, rhs

    def is_zero(self):  # lib.rs:567-573
        return self.kind == "b" and self.val == 0

    def is_one(self):  # lib.rs:578-584
        return self.kind == "b" and self.val == 1

    def key(self):
        if self.kind == "op":
            return ("op", self.op, self.lhs.id, self.rhs.id)
        return (self.kind, self.val)


# Input indicators (lib.rs:133-268) as tuples: ('main'|'aux', column) for single rows,
# ('cm'|'ca'|'nm'|'na', column) for dual rows.
def Main(i):
    return ("main", i)


def Aux(i):
    return ("aux", i)


def CurrentMain(i):
    return ("cm", i)


def CurrentAux(i):
    return ("ca", i)


def NextMain(i):
    return ("nm", i)


def NextAux(i):
    return ("na", i)


def indicator_is_main(ind):
    return ind[0] in ("main", "cm", "nm")


class Builder:
    """ConstraintCircuitBuilder (lib.rs:1013-1148).  `dual` selects DualRowIndicator for new variables."""

    def __init__(self, dual=False):
        self.id_counter = 0
        self.all_nodes = {}   # id -> Node
        self.by_key = {}      # structural key -> Node (equivalent to the reference's linear search)
        self.dual = dual

    # -- leaves (lib.rs:1084-1111) ---------------------------------------------------------------
    def _make_leaf(self, kind, val):
        if kind == "x" and val[1] == 0 and val[2] == 0:  # don't use X field if the B field suffices
            kind, val = "b", val[0]
        key = (kind, val)
        hit = self.by_key.get(key)
        if hit is not None:
            return Monad(hit, self)
        n = Node(self.id_counter, kind, val)
        self.all_nodes[n.id] = n
        self.by_key[key] = n
        self.id_counter += 1
        return Monad(n, self)

    def b_constant(self, v):
        return self._make_leaf("b", v % P)

    def x_constant(self, v):
        if isinstance(v, int):
            v = (v, 0, 0)
        return self._make_leaf("x", tuple(c % P for c in v))

    def zero(self):
        return self.b_constant(0)

    def one(self):
        return self.b_constant(1)

    def minus_one(self):
        return self.b_constant(-1)

    def input(self, indicator):
        return self._make_leaf("in", indicator)

    def challenge(self, c):
        return self._make_leaf("ch", int(c))

    # -- binop (lib.rs:681-740) --------------------------------------------------------------------
    def binop(self, op, lhs, rhs):
        assert lhs.builder is self and rhs.builder is self
        l, r = lhs.node, rhs.node
        if op == ADD and r.is_zero():
            return lhs
        if op == ADD and l.is_zero():
            return rhs
        if op == MUL and r.is_one():
            return lhs
        if op == MUL and l.is_one():
            return rhs
        if op == MUL and r.is_zero():
            return rhs
        if op == MUL and l.is_zero():
            return lhs
        if l.kind in ("b", "x") and r.kind in ("b", "x"):
            if l.kind == "b" and r.kind == "b":
                return self.b_constant(l.val + r.val if op == ADD else l.val * r.val)
            lv = l.val if l.kind == "x" else (l.val, 0, 0)
            rv = r.val if r.kind == "x" else (r.val, 0, 0)
            if op == ADD:
                return self.x_constant(tuple(a + b for a, b in zip(lv, rv)))
            return self.x_constant(xfe_mul(lv, rv))
        # all BinOps are commutative: try both operand orders (swapped first, lib.rs:721-729)
        hit = self.by_key.get(("op", op, r.id, l.id))
        if hit is None:
            hit = self.by_key.get(("op", op, l.id, r.id))
        if hit is not None:
            return Monad(hit, self)
        n = Node(self.id_counter, "op", None, op, l, r)
        self.all_nodes[n.id] = n
        self.by_key[n.key()] = n
        self.id_counter += 1
        return Monad(n, self)

    # lib.rs:1117-1132
    def redirect_all_references_to_node(self, old_id, new_node):
        old = self.all_nodes.pop(old_id)
        if self.by_key.get(old.key()) is old:
            del self.by_key[old.key()]
        for n in list(self.all_nodes.values()):
            if n.kind != "op":
                continue
            if n.lhs.id == old_id or n.rhs.id == old_id:
                if self.by_key.get(n.key()) is n:
                    del self.by_key[n.key()]
                if n.lhs.id == old_id:
                    n.lhs = new_node
                if n.rhs.id == old_id:
                    n.rhs = new_node
                assert n.key() not in self.by_key, "substitution produced a structural duplicate"
                self.by_key[n.key()] = n


class Monad:
    """ConstraintCircuitMonad (lib.rs:635-638): a node plus its builder, with operator overloads
    (lib.rs:742-773).  Evaluation order of operands is left to right in both Rust and Python, which
    is what fixes node ids."""
    __slots__ = ("node", "builder")

    def __init__(self, node, builder):
        self.node, self.builder = node, builder

    def clone(self):
        return self

    def __add__(self, o):
        return self.builder.binop(ADD, self, o)

    def __mul__(self, o):
        return self.builder.binop(MUL, self, o)

    def __neg__(self):
        return self.builder.binop(MUL, self.builder.minus_one(), self)

    def __sub__(self, o):
        return self.builder.binop(ADD, self, -o)


def circuit_sum(iterable):
    """impl Sum (lib.rs:766-773): reduce with +, lazily over the iterator."""
    it = iter(iterable)
    acc = next(it)
    for x in it:
        acc = acc + x
    return acc


def circuit_product(iterable):
    it = iter(iterable)
    acc = next(it)
    for x in it:
        acc = acc * x
    return acc


# ---------------------------------------------------------------------------------------------
# degrees and traversals
def degrees(roots):
    """degree() (lib.rs:515-541) for every node reachable from `roots`, by node id."""
    deg = {}

    def go(n):
        d = deg.get(n.id)
        if d is not None:
            return d
        if n.kind == "op":
            dl, dr = go(n.lhs), go(n.rhs)
            if n.op == ADD:
                d = max(dl, dr)
            else:
                d = -1 if min(dl, dr) <= -1 else dl + dr
        elif n.kind == "in":
            d = 1
        else:
            d = -1 if n.is_zero() else 0
        deg[n.id] = d
        return d

    # iterative post-order to be safe with deep chains
    for r in roots:
        stack = [(r, False)]
        while stack:
            n, done = stack.pop()
            if n.id in deg:
                continue
            if n.kind != "op":
                go(n)
                continue
            if done:
                go(n)
            else:
                stack.append((n, True))
                if n.rhs.id not in deg:
                    stack.append((n.rhs, False))
                if n.lhs.id not in deg:
                    stack.append((n.lhs, False))
    return deg


def reachable(roots):
    """All distinct nodes reachable from roots, in a topological order (children before parents)."""
    seen, order = set(), []
    for r in roots:
        stack = [(r, False)]
        while stack:
            n, done = stack.pop()
            if done:
                order.append(n)
                continue
            if id(n) in seen:
                continue
            seen.add(id(n))
            stack.append((n, True))
            if n.kind == "op":
                stack.append((n.rhs, False))
                stack.append((n.lhs, False))
    return order


def evaluates_to_base_element(n, memo=None):
    """lib.rs:598-609"""
    memo = {} if memo is None else memo
    order = reachable([n])
    for m in order:
        if m.id in memo:
            continue
        if m.kind == "b":
            memo[m.id] = True
        elif m.kind in ("x", "ch"):
            memo[m.id] = False
        elif m.kind == "in":
            memo[m.id] = indicator_is_main(m.val)
        else:
            memo[m.id] = memo[m.lhs.id] and memo[m.rhs.id]
    return memo[n.id]


def num_visible_nodes(roots):
    """lib.rs:988-996"""
    return len(reachable([r.node if isinstance(r, Monad) else r for r in roots]))


def multicircuit_degree(roots):
    deg = degrees(roots)
    return max((deg[r.id] for r in roots), default=-1)


# ---------------------------------------------------------------------------------------------
# degree lowering (lib.rs:820-985)
def pick_node_to_substitute(roots, target_degree):
    """lib.rs:902-967.  The reference counts how often every low-degree node occurs in the *tree
    expansions* of all distinct high-degree nodes; that multiset count equals the number of DAG paths
    from high-degree nodes, computed here by dynamic programming."""
    deg = degrees(roots)
    order = reachable(roots)                      # children before parents
    high = {n.id for n in order if deg[n.id] > target_degree}
    # T[x] = sum over high nodes H of (number of paths H -> x); start with the empty path for H itself
    T = {n.id: (1 if n.id in high else 0) for n in order}
    for n in reversed(order):                     # parents before children
        t = T[n.id]
        if t and n.kind == "op":
            T[n.lhs.id] += t
            T[n.rhs.id] += t
    cand = {n.id: T[n.id] for n in order if T[n.id] > 0 and 1 < deg[n.id] <= target_degree}
    assert cand, "Cannot lower degree."
    max_occ = max(cand.values())
    ids = [i for i, c in cand.items() if c == max_occ]
    max_deg = max(deg[i] for i in ids)
    ids = [i for i in ids if deg[i] == max_deg]
    return min(ids)


def lower_to_degree(multicircuit, builder, target_degree, num_main_cols, num_aux_cols):
    """lib.rs:820-869 + apply_substitution (lib.rs:871-897).  `multicircuit` is a list of Monads,
    modified in place; returns (main_constraints, aux_constraints) as lists of Monads."""
    assert target_degree > 1
    main_constraints, aux_constraints = [], []
    if not multicircuit:
        return main_constraints, aux_constraints
    # the roots are re-pointed in place below: give every slot its own wrapper object
    multicircuit[:] = [Monad(m.node, builder) for m in multicircuit]
    while multicircuit_degree([m.node for m in multicircuit]) > target_degree:
        chosen_id = pick_node_to_substitute([m.node for m in multicircuit], target_degree)
        chosen = builder.all_nodes[chosen_id]
        is_main = evaluates_to_base_element(chosen)
        if is_main:
            idx = num_main_cols + len(main_constraints)
            indicator = CurrentMain(idx) if builder.dual else Main(idx)
        else:
            idx = num_aux_cols + len(aux_constraints)
            indicator = CurrentAux(idx) if builder.dual else Aux(idx)
        new_variable = builder.input(indicator)
        builder.redirect_all_references_to_node(chosen_id, new_variable.node)
        for m in multicircuit:
            if m.node.id == chosen_id:
                m.node = new_variable.node
        new_constraint = new_variable - Monad(chosen, builder)
        if evaluates_to_base_element(new_constraint.node):
            main_constraints.append(new_constraint)
        else:
            aux_constraints.append(new_constraint)
    return main_constraints, aux_constraints
