"""Pin of the whole AIR restatement (tools/air -> oracle/air_circuit.h -> csrc/air_gen_*.hip) against the
REFERENCE's own golden value: `air_constraints_evaluators_have_not_changed`
(/root/reference/triton-vm/src/table/master_table.rs:2328-2414).

The reference seeds `StdRng::seed_from_u64(3508729174085202315)`, draws pseudorandom current/next main rows (once as
BFieldElements, once as XFieldElements), auxiliary rows and 63 challenges, evaluates all four generated evaluators on
both kinds of main row, reads the 2 x 604 results as the coefficients of one polynomial and evaluates it at one more
random XFieldElement.  One XFieldElement fingerprints every constant, sign, column index and challenge id of all 604
constraints (and their order): it cannot match by accident.

The generator (rand / twenty-first are not vendored) is restated in oracle/ref_rng.py; the variant that reproduces
the reference value -- ChaCha12, PCG32 seed expansion, rand 0.9-style widening-multiply range sampling of
`0..=BFieldElement::MAX` -- is recorded here, and the others are asserted NOT to match.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import ref_rng as rr

SEED = 3508729174085202315
EXPECTED = [17974882881108171077, 15638927082579294872, 9717283721935042729]   # master_table.rs:2401-2405
SECTION_ENDS = [0, 81, 178, 581, 604]


def reference_inputs(expand=rr.pcg32_seed, sampler=rr.StdRng.range_canon):
    """The draws of master_table.rs:2329-2341, in order (canonical values)."""
    rng = rr.StdRng.seed_from_u64(SEED, expand=expand)
    b = lambda: sampler(rng)
    bfes = lambda n: [b() for _ in range(n)]
    xfes = lambda n: [[b(), b(), b()] for _ in range(n)]
    d = dict(main_cur_base=bfes(379), main_cur_ext=xfes(379), aux_cur=xfes(91),
             main_next_base=bfes(379), main_next_ext=xfes(379), aux_next=xfes(91), challenges=xfes(63))
    return d, (lambda: [b(), b(), b()])


def fingerprint(evaluator, **kw):
    d, draw_point = reference_inputs(**kw)
    m = {k: orc.to_mont(v) for k, v in d.items()}
    base = evaluator(m["main_cur_base"], m["main_next_base"], m["aux_cur"], m["aux_next"], m["challenges"])
    ext = evaluator(m["main_cur_ext"], m["main_next_ext"], m["aux_cur"], m["aux_next"], m["challenges"])
    parts = []
    for s in range(4):                                                   # master_table.rs:2385-2396
        parts += [base[SECTION_ENDS[s]:SECTION_ENDS[s + 1]], ext[SECTION_ENDS[s]:SECTION_ENDS[s + 1]]]
    coefficients = np.concatenate(parts)
    point = orc.to_mont(draw_point())
    return [int(v) for v in orc.from_mont(orc.poly_eval_xfe(coefficients, point))]


def test_oracle_circuit_reproduces_the_reference_fingerprint():
    assert fingerprint(orc.air_constraint_values) == EXPECTED


@pytest.mark.parametrize("sampler", ["zone", "reject", "mod"])
def test_other_sampler_variants_do_not_match(sampler):
    assert fingerprint(orc.air_constraint_values, sampler=rr.BFE_SAMPLERS[sampler]) != EXPECTED


def test_python_circuit_reproduces_the_reference_fingerprint():
    """The same value from the Python-side circuit objects of tools/air (the generator input itself), walked
    independently of the exported C tables."""
    from tools.air import build
    from tools.air.circuit import reachable

    P = rr.P

    def xmul(a, b):
        a0, a1, a2 = a
        b0, b1, b2 = b
        c0, c1, c2, c3, c4 = a0 * b0, a0 * b1 + a1 * b0, a0 * b2 + a1 * b1 + a2 * b0, a1 * b2 + a2 * b1, a2 * b2
        # X^3 = X - 1, X^4 = X^2 - X
        return ((c0 - c3) % P, (c1 + c3 - c4) % P, (c2 + c4) % P)

    low, _, _, _ = build.lower(build.all_constraints()[0])

    def evaluator(mc, mn, ac, an, ch):
        lift = lambda v: tuple(int(t) for t in v) if np.ndim(v) else (int(v), 0, 0)
        fm = lambda a: orc.from_mont(a)
        mc, mn, ac, an, ch = fm(mc), fm(mn), fm(ac), fm(an), fm(ch)
        out = []
        for sec in ("init", "cons", "tran", "term"):
            _, roots = low[sec]
            base, ext = build.ordered_roots(roots)
            val = {}
            for n in reachable(base + ext):
                if n.kind == "op":
                    l, r = val[id(n.lhs)], val[id(n.rhs)]
                    val[id(n)] = tuple((x + y) % P for x, y in zip(l, r)) if n.op == "+" else xmul(l, r)
                else:
                    val[id(n)] = leaf(n, mc, mn, ac, an, ch, lift)
            out += [val[id(n)] for n in base + ext]
        return orc.to_mont(np.array(out, dtype=object))

    assert fingerprint(evaluator) == EXPECTED


def leaf(n, mc, mn, ac, an, ch, lift):
    k, v = n.kind, n.val
    if k == "b":
        return (int(v) % rr.P, 0, 0)
    if k == "x":
        return tuple(int(t) % rr.P for t in v)
    if k == "ch":
        return lift(ch[v])
    if k == "in":
        kind, idx = v
        return lift({"main": mc, "cm": mc, "nm": mn, "aux": ac, "ca": ac, "na": an}[kind][idx])
    raise AssertionError(k)
