"""The degree-lowering fill (tvm_fill_derived_main_columns / tvm_fill_derived_aux_columns) against its oracle, and
the structural pin: on the filled tables every substitution constraint `x - expr` of the lowered AIR vanishes."""
import numpy as np
import pytest

from oracle import degree_lowering as dlo
from oracle.degree_lowering_rules import RULES
from triton_vm_amd import degree_lowering as dl


def _tables(orc, rng, n, aux_cols=91):
    main = orc.random_elements(rng, (379, n))
    main[dl.FIRST_DERIVED_MAIN:] = 0
    aux = orc.random_elements(rng, (aux_cols, n, 3))
    aux[dl.FIRST_DERIVED_AUX:dl.NUM_AUX_COLUMNS] = 0
    challenges = orc.random_elements(rng, (63, 3))
    return main, aux, challenges


def test_rule_table_shape():
    assert [(s, t, c, len(e)) for s, t, c, e in RULES] == [("init", "main", 149, 2), ("cons", "main", 151, 18),
                                                           ("tran", "main", 169, 210), ("tran", "aux", 49, 41)]


@pytest.mark.parametrize("n", [2, 8, 64, 300])
def test_fill_matches_oracle(ctx, orc, n):
    rng = np.random.default_rng(n)
    main, aux, challenges = _tables(orc, rng, n)
    d_main, d_aux = ctx.to_device(main), ctx.to_device(aux)
    dl.fill_derived_main_columns(ctx, d_main, n)
    dl.fill_derived_aux_columns(ctx, d_main, d_aux, n, challenges)
    got_main, got_aux = d_main.download((379, n)), d_aux.download((91, n, 3))
    want_main, want_aux = dlo.fill(main, aux, challenges)
    assert (got_main[:dl.FIRST_DERIVED_MAIN] == main[:dl.FIRST_DERIVED_MAIN]).all()       # inputs untouched
    assert (got_main == want_main).all()
    assert (got_aux == want_aux).all()
    assert (got_aux[90] == aux[90]).all()                                                 # the randomizer column too
    assert (got_main[169:, n - 1] == 0).all() and (got_aux[49:90, n - 1] == 0).all()     # transition columns, last row


def test_substitution_constraints_vanish_on_the_filled_tables(ctx, orc):
    """x - expr = 0 for every derived column, evaluated independently of the fill order: each rule on the FINAL
    tables (a rule that read a not-yet-derived column, or the wrong row, would show up here)."""
    n = 16
    rng = np.random.default_rng(5)
    main, aux, challenges = _tables(orc, rng, n)
    d_main, d_aux = ctx.to_device(main), ctx.to_device(aux)
    dl.fill_derived_main_columns(ctx, d_main, n)
    dl.fill_derived_aux_columns(ctx, d_main, d_aux, n, challenges)
    m = [[int(v) * dlo.R_INV % dlo.P for v in col] for col in d_main.download((379, n))]
    x = [[tuple(int(w) * dlo.R_INV % dlo.P for w in el) for el in col] for col in d_aux.download((91, n, 3))]
    ch = [tuple(int(w) * dlo.R_INV % dlo.P for w in c) for c in challenges]
    for sec, table, start, exprs in RULES:
        for r in range(n - 1 if sec == "tran" else n):
            env = {"MC": lambda c: m[c][r], "MN": lambda c: m[c][r + 1], "ch": lambda k: ch[k],
                   "AC": lambda c: x[c][r], "AN": lambda c: x[c][r + 1]}
            for k, e in enumerate(exprs):
                v = dlo.evaluate(e, env)
                have = m[start + k][r] if table == "main" else x[start + k][r]
                assert dlo.lift(v) == dlo.lift(have), (sec, table, start + k, r)


def test_argument_checks(ctx):
    from triton_vm_amd.capi import TritonHipError

    d = ctx.alloc(379 * 4)
    with pytest.raises(TritonHipError):
        dl.fill_derived_main_columns(ctx, d, 1)           # a transition needs two rows
    with pytest.raises(ValueError):
        dl.fill_derived_main_columns(ctx, ctx.alloc(10), 4)
