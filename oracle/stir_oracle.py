"""TEST INFRASTRUCTURE ONLY -- a CPU restatement of the arithmetic of one STIR prover round
(/root/reference/triton-vm/src/low_degree_test/stir.rs:885-993), written the way the reference writes it: with
polynomials in coefficient form, Lagrange interpolation, an explicit zerofier, polynomial long division and
schoolbook multiplication (twenty-first's Polynomial::{interpolate, zerofier, /, *}).  Pure-Python loops over the C
oracle's field arithmetic.  The product never imports this file.  PARITY: no reference-held STIR vector exists (its own tests are
prove-then-verify); since round 6 oracle/real_prover.py builds whole STIR proofs from this file and the device proofs equal them --
DESIGN.md section 7.
"""
import numpy as np

from . import oracle as orc

ZERO = np.zeros(3, np.uint64)


def one():
    return np.array([orc.bfe(1), 0, 0], np.uint64)


def lift(b):
    return np.array([b, 0, 0], np.uint64)


def xfe_neg(a):
    return orc.xfe_sub(ZERO, a)


def poly_trim(p):
    p = list(p)
    while p and not np.any(p[-1]):
        p.pop()
    return p


def poly_add(a, b):
    n = max(len(a), len(b))
    return [orc.xfe_add(a[i] if i < len(a) else ZERO, b[i] if i < len(b) else ZERO) for i in range(n)]


def poly_sub(a, b):
    n = max(len(a), len(b))
    return [orc.xfe_sub(a[i] if i < len(a) else ZERO, b[i] if i < len(b) else ZERO) for i in range(n)]


def poly_mul(a, b):
    if not a or not b:
        return []
    out = [ZERO.copy() for _ in range(len(a) + len(b) - 1)]
    for i, x in enumerate(a):
        if not np.any(x):
            continue
        for j, y in enumerate(b):
            out[i + j] = orc.xfe_add(out[i + j], orc.xfe_mul(x, y))
    return out


def poly_divmod(num, den):
    """schoolbook long division; den non-zero"""
    num, den = poly_trim(num), poly_trim(den)
    lead_inv = orc.xfe_inv(den[-1])
    quot = [ZERO.copy() for _ in range(max(len(num) - len(den) + 1, 0))]
    rem = [c.copy() for c in num]
    for i in range(len(num) - len(den), -1, -1):
        q = orc.xfe_mul(rem[i + len(den) - 1], lead_inv)
        quot[i] = q
        for j, d in enumerate(den):
            rem[i + j] = orc.xfe_sub(rem[i + j], orc.xfe_mul(q, d))
    return quot, poly_trim(rem)


def zerofier(points):
    """prod (X - p)"""
    z = [one()]
    for p in points:
        z = poly_mul(z, [xfe_neg(p), one()])
    return z


def lagrange_interpolate(points, values):
    """the polynomial of degree < k through (points[i], values[i])"""
    k = len(points)
    total = []
    for i in range(k):
        others = [points[j] for j in range(k) if j != i]
        basis = zerofier(others)
        denom = one()
        for p in others:
            denom = orc.xfe_mul(denom, orc.xfe_sub(points[i], p))
        scale = orc.xfe_mul(values[i], orc.xfe_inv(denom))
        total = poly_add(total, [orc.xfe_mul(c, scale) for c in basis])
    return total


def divide_by_linear(poly, root):
    """poly / (X - root) for a root of poly: synthetic division (the quotient's coefficients, lowest first)"""
    out = [ZERO.copy() for _ in range(len(poly) - 1)]
    carry = ZERO.copy()
    for i in range(len(poly) - 1, 0, -1):
        carry = orc.xfe_add(poly[i], orc.xfe_mul(carry, root))
        out[i - 1] = carry
    return out


def lagrange_interpolate_from_zerofier(points, values):
    """the same polynomial as lagrange_interpolate, in O(k^2): the basis polynomial of point i is Z / (X - p_i) over its own value at
    p_i, with Z the zerofier of all points (what lets the oracle prover handle the ~200 points of a STIR round in seconds; the cubic
    form above stays for the small cases, and the two are compared in tests/test_stir.py)"""
    k = len(points)
    z = zerofier(points)
    total = [ZERO.copy() for _ in range(k)]
    for i in range(k):
        basis = divide_by_linear(z, points[i])
        at_point = orc.poly_eval_xfe(np.array(basis, np.uint64), points[i])
        scale = orc.xfe_mul(values[i], orc.xfe_inv(at_point))
        total = poly_add(total, [orc.xfe_mul(c, scale) for c in basis])
    return total


def fold_polynomial(coeffs, folding_factor, randomness):
    """stir.rs:1132-1147: every chunk of `folding_factor` coefficients evaluated at the randomness"""
    coeffs = np.asarray(coeffs, np.uint64).reshape(-1, 3)
    return np.array([orc.poly_eval_xfe(coeffs[i:i + folding_factor], randomness)
                     for i in range(0, len(coeffs), folding_factor)], np.uint64).reshape(-1, 3)


def stack(codeword, stack_height):
    """stir.rs:1402-1419"""
    codeword = np.asarray(codeword, np.uint64).reshape(-1, 3)
    distance = -(-len(codeword) // stack_height)
    return [codeword[skip::distance] for skip in range(distance)]


def stir_merkle_root(codeword, stack_height):
    """StirMerkleTree::new (stir.rs:1380-1400) -> root digest"""
    digests = np.array([orc.hash_varlen(np.ascontiguousarray(s).reshape(-1)) for s in stack(codeword, stack_height)], np.uint64)
    return orc.merkle_tree(digests)[1]


def next_polynomial(folded, quotient_set, quotient_answers, degree_correction_randomness, interpolate=None):
    """stir.rs:945-966: ((folded - Ans) / Zerofier) * (1 + r X + ... + r^k X^k), coefficients"""
    folded = [c for c in np.asarray(folded, np.uint64).reshape(-1, 3)]
    points = [p for p in np.asarray(quotient_set, np.uint64).reshape(-1, 3)]
    answers = [v for v in np.asarray(quotient_answers, np.uint64).reshape(-1, 3)]
    ans = (interpolate or lagrange_interpolate)(points, answers)
    quotient, remainder = poly_divmod(poly_sub(folded, ans), zerofier(points))
    assert not remainder, "folded - Ans must vanish on the quotient set"
    correction, power = [], one()
    for _ in range(len(points) + 1):
        correction.append(power)
        power = orc.xfe_mul(power, degree_correction_randomness)
    return poly_mul(quotient, correction)


def stack_tree(codeword, stack_height):
    """StirMerkleTree::new (stir.rs:1380-1400) -> (the stacks, the node array of the tree over their digests)"""
    stacks = stack(codeword, stack_height)
    digests = np.array([orc.hash_varlen(np.ascontiguousarray(s).reshape(-1)) for s in stacks], np.uint64)
    return stacks, orc.merkle_tree(digests)
