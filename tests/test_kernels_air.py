"""csrc/air.hip (scheduled AIR program, LDS slot file) against the oracle's literal walk of the same
lowered circuit DAG, through the C ABI: on the CPU fiber emulation (-m "not gpu") and on the MI355X (-m gpu).
Tables are random (the AIR is just a polynomial map here), full width: 379 main / 91 aux columns."""
import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, MasterTable, field, stark


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


@pytest.mark.parametrize("log_n,expansion,ldt_expansion", [(2, 8, 8), (3, 4, 4), (3, 4, 8), (4, 1, 1), (4, 2, 2)])  # 1, 2: per-rank domains of the sharded prover
def test_all_quotients_combined(ctx, orc, log_n, expansion, ldt_expansion):
    rng = np.random.default_rng(log_n * 10 + expansion + ldt_expansion)
    n, h = 1 << log_n, 3
    g = field.generator()
    trace_dom = ArithmeticDomain.of_length(n)
    quot = ArithmeticDomain.of_length(n * expansion).with_offset(g)
    ldt = ArithmeticDomain.of_length(n * ldt_expansion).with_offset(g)
    main = MasterTable(ctx, orc.random_elements(rng, (379, n)), orc.random_elements(rng, (379, h)), trace_dom, quot, ldt, 1)
    aux = MasterTable(ctx, orc.random_elements(rng, (91, n, 3)), orc.random_elements(rng, (91, h, 3)), trace_dom, quot, ldt, 3)
    main.maybe_low_degree_extend_all_columns()
    aux.maybe_low_degree_extend_all_columns()
    challenges = orc.random_elements(rng, (63, 3))
    weights = orc.random_elements(rng, (604, 3))
    got = stark.all_quotients_combined(ctx, main, aux, trace_dom, quot, challenges, weights).download((len(quot), 3))

    stride = len(main.evaluation_domain()) // len(quot)
    main_rows = np.ascontiguousarray(main.low_degree_extended_table()[::stride])
    aux_rows = np.ascontiguousarray(aux.low_degree_extended_table()[::stride])
    want = orc.quotients_combined(main_rows, aux_rows, odom(orc, trace_dom), odom(orc, quot), challenges, weights)
    assert (got == want).all()
