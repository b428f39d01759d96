"""csrc/air.hip (scheduled AIR program, LDS slot file) against the oracle's literal walk of the same
lowered circuit DAG, through the C ABI: on the CPU fiber emulation (-m "not gpu") and on the MI355X (-m gpu).
Tables are random (the AIR is just a polynomial map here), full width: 379 main / 91 aux columns."""
import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, MasterTable, field, stark


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


@pytest.mark.parametrize("log_n,expansion,ldt_expansion", [
    (2, 8, 8), (3, 4, 4), (3, 4, 8), (4, 1, 1), (4, 2, 2),  # 1, 2: per-rank domains of the sharded prover
    # several 256-row workgroups (block-base addressing, wrap rows of the last block); GPU only: the emulation needs minutes
    pytest.param(8, 8, 8, marks=pytest.mark.gpu), pytest.param(9, 4, 16, marks=pytest.mark.gpu)])
def test_all_quotients_combined(ctx, orc, log_n, expansion, ldt_expansion):
    if log_n >= 8 and ctx.kind != "gpu":
        pytest.skip("multi-workgroup case runs on the MI355X")
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


def _python_quotient(orc, main_cur, main_next, aux_cur, aux_next, challenges, weights, x, n_trace, omega_inv):
    """all_quotients_combined for ONE row (master_table.rs:1302-1359) in python integers: constraint values from the
    oracle's DAG walk, zerofier inverses (master_table.rs:1194-1250) and the weighted sums recomputed here."""
    P = orc.P
    vals = orc.from_mont(orc.air_constraint_values(main_cur, main_next, aux_cur, aux_next, challenges)).astype(object)
    w = orc.from_mont(weights).astype(object)

    def xmul(a, b):
        c0, c1, c2 = a[0] * b[0], a[0] * b[1] + a[1] * b[0], a[0] * b[2] + a[1] * b[1] + a[2] * b[0]
        c3, c4 = a[1] * b[2] + a[2] * b[1], a[2] * b[2]
        return [(c0 - c3) % P, (c1 + c3 - c4) % P, (c2 + c4) % P]

    xn = pow(x, n_trace, P)
    z_init = pow(x - 1, -1, P)
    z_cons = pow(xn - 1, -1, P)
    z_tran = (x - omega_inv) * z_cons % P
    z_term = pow(x - omega_inv, -1, P)
    ends = [0, 81, 178, 581, 604]
    q = [0, 0, 0]
    for s, z in enumerate((z_init, z_cons, z_tran, z_term)):
        acc = [0, 0, 0]
        for k in range(ends[s], ends[s + 1]):
            t = xmul(vals[k], w[k])
            acc = [(a + b) % P for a, b in zip(acc, t)]
        q = [(a + z * b) % P for a, b in zip(q, acc)]
    return q



def check_sampled_quotient_rows(ctx, orc, log_n, log_ldt_expansion, h, synthetic, n_random=24):
    """Sampled-row parity of all_quotients_combined against the oracle's DAG walk (rows i and i + |Q|/N revealed from
    the extended tables); quotient domain 8N, LDT domain N << log_ldt_expansion."""
    from triton_vm_amd import stark

    n = 1 << log_n
    g = field.generator()
    trace_dom = ArithmeticDomain.of_length(n)
    quot = ArithmeticDomain.of_length(8 * n).with_offset(g)
    ldt = ArithmeticDomain.of_length(n << log_ldt_expansion).with_offset(g)
    tables = []
    for fk, n_cols in ((1, 379), (3, 91)):
        mt = MasterTable.__new__(MasterTable)
        mt.ctx, mt.fk, mt.n_cols, mt.n_rows, mt.num_trace_randomizers = ctx, fk, n_cols, n, h
        mt.trace_domain, mt.quotient_domain, mt.ldt_domain, mt._table = trace_dom, quot, ldt, None
        shape = lambda k: (n_cols, k) + ((3,) if fk == 3 else ())
        if synthetic:        # filled on the device (full-size tables)
            mt.d_trace = ctx.synthetic(n_cols * n * fk, seed=21 + fk)
            mt.d_randomizers = ctx.synthetic(n_cols * h * fk, seed=23 + fk)
        else:
            r = np.random.default_rng(fk)
            mt.d_trace = ctx.to_device(orc.random_elements(r, shape(n)))
            mt.d_randomizers = ctx.to_device(orc.random_elements(r, shape(h)))
        mt.maybe_low_degree_extend_all_columns()
        tables.append(mt)
    main, aux = tables
    rng = np.random.default_rng(77)
    challenges = orc.random_elements(rng, (63, 3))
    weights = orc.random_elements(rng, (604, 3))
    d_q = stark.all_quotients_combined(ctx, main, aux, trace_dom, quot, challenges, weights)

    Q, unit = len(quot), len(quot) // n
    stride = len(main.evaluation_domain()) // Q          # quotient-domain row i is evaluation-domain row i * stride
    sample = np.unique(np.concatenate([
        np.array([0, 1, 255, 256, 257, Q // 2 - 1, Q // 2, Q - 257, Q - 256]) % Q, np.arange(max(Q - unit - 2, 0), Q),
        rng.integers(0, Q, n_random), rng.integers(3 * Q // 4, Q, n_random)])).astype(np.uint64)
    nxt = (sample + np.uint64(unit)) % np.uint64(Q)
    got = np.empty((sample.size, 3), np.uint64)
    ctx._check(ctx.lib.tvm_gather_elements(ctx.handle, d_q.ptr, 3, sample.ctypes.data, sample.size, got.ctypes.data), "gather")
    rows_m = main.reveal_rows(np.concatenate([sample, nxt]) * np.uint64(stride))
    rows_a = aux.reveal_rows(np.concatenate([sample, nxt]) * np.uint64(stride))
    k = sample.size
    P = orc.P
    gen, off = orc.value(quot.generator), orc.value(quot.offset)
    omega_inv = pow(orc.value(trace_dom.generator), -1, P)
    for j, i in enumerate(sample):
        x = off * pow(gen, int(i), P) % P
        want = _python_quotient(orc, rows_m[j], rows_m[k + j], rows_a[j], rows_a[k + j], challenges, weights, x, n, omega_inv)
        assert [int(v) for v in orc.from_mont(got[j])] == want, f"quotient row {i}"
    main.clear_cache()
    aux.clear_cache()


@pytest.mark.parametrize("log_n,log_ldt_expansion", [(3, 3), (2, 5)])
def test_sampled_quotient_rows_small(ctx, orc, log_n, log_ldt_expansion):
    """The sampled-row checker of tests/test_gpu_fullsize.py (python-integer zerofiers and weighted sums over the
    oracle's constraint values) at a size where every row is sampled, incl. the stride-4 view."""
    check_sampled_quotient_rows(ctx, orc, log_n, log_ldt_expansion, 3, synthetic=False, n_random=64)


def _quotient(ctx, orc, main_trace, aux_trace, h, seed, ch=None, valid_mode=False, fork=0):
    """fork: TVM_OPTION_AIR_FORK_MAX_WORKGROUPS for the call -- 0 keeps the parts on one stream and lets valid-trace mode split
    these short domains (the library's default, 256, would evaluate them row by row on the fork lanes)"""
    rng = np.random.default_rng(seed)
    n = main_trace.shape[1]
    g = field.generator()
    trace_dom = ArithmeticDomain.of_length(n)
    quot = ArithmeticDomain.of_length(8 * n).with_offset(g)
    main = MasterTable(ctx, main_trace, orc.random_elements(rng, (379, h)), trace_dom, quot, quot, 1)
    aux = MasterTable(ctx, aux_trace, orc.random_elements(rng, (91, h, 3)), trace_dom, quot, quot, 3)
    main.maybe_low_degree_extend_all_columns()
    aux.maybe_low_degree_extend_all_columns()
    challenges = orc.random_elements(rng, (63, 3)) if ch is None else ch
    weights = orc.random_elements(rng, (604, 3))
    ctx.assume_valid_trace(valid_mode)
    ctx.air_fork_max_workgroups(fork)
    try:
        got = stark.all_quotients_combined(ctx, main, aux, trace_dom, quot, challenges, weights).download((len(quot), 3))
    finally:
        ctx.assume_valid_trace(False)
        ctx.air_fork_max_workgroups(256)
    want = orc.quotients_combined(main.low_degree_extended_table(), aux.low_degree_extended_table(), odom(orc, trace_dom),
                                  odom(orc, quot), challenges, weights)
    return got, want, quot


def test_valid_trace_mode_is_exact_on_a_valid_trace(ctx, orc):
    """TVM_OPTION_AIR_VALID_TRACE: consistency / transition constraints on half of the quotient domain + interpolation.
    On the valid 256-row trace of a real execution (tests/vm_fixture.py) the quotient codeword equals the oracle's
    row-by-row evaluation bit for bit -- and it is a polynomial of degree < 4 (N + h)."""
    from tests import vm_fixture as vf

    main_trace, aux_trace, ch, _ = vf.valid_tables("tiny")
    h = 5
    got, want, quot = _quotient(ctx, orc, main_trace, aux_trace, h, 3, ch=ch, valid_mode=True)
    assert (got == want).all()
    coeffs = quot.interpolate(ctx, ctx.to_device(got), 3).download((len(quot), 3))
    assert not coeffs[4 * (256 + h):].any() and coeffs[:4 * 256].any()


def test_forked_parts_give_the_same_words(ctx, orc):
    """The parts of the AIR on four streams, one accumulator per lane, summed by the scatter (the default on short quotient domains)
    against the parts one behind another on the context's stream: the same codeword, and the oracle's.  In valid-trace mode a
    domain short enough to fork is evaluated row by row: exact on ANY trace, so the random tables below come out as the oracle's."""
    rng = np.random.default_rng(21)
    n, h = 16, 3
    main_trace, aux_trace = orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))
    serial, want, _ = _quotient(ctx, orc, main_trace, aux_trace, h, 5, fork=0)
    forked, _, _ = _quotient(ctx, orc, main_trace, aux_trace, h, 5, fork=256)
    assert (serial == want).all() and (forked == want).all()
    forked_valid, _, _ = _quotient(ctx, orc, main_trace, aux_trace, h, 5, valid_mode=True, fork=256)
    assert (forked_valid == want).all()


@pytest.mark.gpu
def test_valid_trace_classes_on_the_fork_lanes(ctx, orc):
    """Valid-trace mode whose class evaluations fork: the quotient domain of the 256-row trace is 8 workgroups -- above a limit of 4
    it is split into the classes, whose half / quarter / single-coset domains (4 / 2 / 1 workgroups) run their parts side by side."""
    if ctx.kind != "gpu":
        pytest.skip("several 256-row workgroups: runs on the MI355X")
    from tests import vm_fixture as vf

    main_trace, aux_trace, ch, _ = vf.valid_tables("tiny")
    got, want, _ = _quotient(ctx, orc, main_trace, aux_trace, 5, 3, ch=ch, valid_mode=True, fork=4)
    assert (got == want).all()


def test_valid_trace_mode_changes_only_interpolated_rows_of_an_invalid_trace(ctx, orc):
    """On random tables the constraint quotients are rational functions: the default (row-by-row) mode reproduces the
    reference's values, the valid-trace mode agrees with it exactly on the rows where it evaluates EVERY constraint (the
    rows 0 mod 4: the constraints of low degree are evaluated on a quarter of the points, the others on half of them) and
    nowhere else -- which is why it is an opt-in whose precondition is a valid trace."""
    rng = np.random.default_rng(9)
    n, h = 16, 3
    main_trace, aux_trace = orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))
    exact, want, _ = _quotient(ctx, orc, main_trace, aux_trace, h, 4)
    assert (exact == want).all()
    fast, _, _ = _quotient(ctx, orc, main_trace, aux_trace, h, 4, valid_mode=True)
    assert (fast[0::4] == want[0::4]).all()
    assert (fast[1::2] != want[1::2]).any(axis=1).all()
    assert (fast[2::4] != want[2::4]).any(axis=1).all()    # (n = 16, h = 3: the quarter-domain class is in use)
