"""Parity at BASELINE.json's full size (config 1: 2^20 padded rows, 379 main / 91 aux columns,
evaluation domain 2^23) on the real MI355X.  The oracle cannot extend 652 columns of 2^23 values in
seconds, so whole columns and whole rows are sampled: sampled columns are extended by the oracle at
full length and compared at sampled rows; sampled rows are re-hashed by the oracle; sampled leaves
are authenticated against the device-built Merkle root with the oracle's hash_pair."""
import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, MasterTable, field

pytestmark = pytest.mark.gpu

LOG_N = 20
H = 198  # trace randomizers with FRI at 160-bit security


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


@pytest.fixture(scope="module")
def gctx():
    from triton_vm_amd import Context

    c = Context(device=0)
    yield c
    c.close()


@pytest.mark.parametrize("fk,n_cols,sample_cols", [(1, 379, (0, 200, 378)), (3, 91, (0, 90))])
def test_full_size_table(gctx, orc, fk, n_cols, sample_cols):
    ctx = gctx
    n = 1 << LOG_N
    trace_dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(8 * n).with_offset(field.generator())
    mt = MasterTable.__new__(MasterTable)
    mt.ctx, mt.fk, mt.n_cols, mt.n_rows, mt.num_trace_randomizers = ctx, fk, n_cols, n, H
    mt.trace_domain, mt.quotient_domain, mt.ldt_domain, mt._table = trace_dom, ev, ev, None
    mt.d_trace = ctx.synthetic(n_cols * n * fk, seed=11 + fk)
    mt.d_randomizers = ctx.synthetic(n_cols * H * fk, seed=13 + fk)
    mt.maybe_low_degree_extend_all_columns()
    ctx.sync()

    rng = np.random.default_rng(5)
    rows = np.unique(np.concatenate([[0, 1, 7, 8, len(ev) - 1], rng.integers(0, len(ev), 500)])).astype(np.uint64)
    revealed = mt.reveal_rows(rows)  # [n_rows_sampled, n_cols(, 3)]

    # whole columns through the oracle
    trace_host = mt.d_trace.download().reshape((n_cols, n) + ((3,) if fk == 3 else ()))
    rnd_host = mt.d_randomizers.download().reshape((n_cols, H) + ((3,) if fk == 3 else ()))
    for c in sample_cols:
        want = orc.lde_table(trace_host[c:c + 1], rnd_host[c:c + 1], odom(orc, ev), fk)
        assert (revealed[:, c] == want[rows.astype(np.int64), 0]).all(), f"column {c}"

    # whole rows through the oracle hash
    nodes = mt.merkle_tree()
    L = len(ev)
    for j, r in enumerate(rows[:64]):
        digest = orc.hash_varlen(revealed[j].reshape(-1))
        assert (nodes[L + int(r)] == digest).all(), f"row {r}"
        # authentication path up to the root
        i = L + int(r)
        cur = digest
        while i > 1:
            sib = nodes[i ^ 1]
            cur = orc.hash_pair(cur, sib) if i % 2 == 0 else orc.hash_pair(sib, cur)
            i >>= 1
            assert (nodes[i] == cur).all()
    mt.clear_cache()


@pytest.mark.parametrize("log_len", [23, 25])
def test_large_xfe_transform_round_trip_and_point_values(gctx, log_len):
    """Size-independent properties of the codeword transforms at the lengths of the 2^20- and 2^22-row
    configurations (quotient domain 2^23 / 2^25): interpolate(evaluate(f)) == f, and the codeword's entry i
    equals f(offset * generator^i) evaluated by the independent Horner kernel."""
    from triton_vm_amd import stark

    n = 1 << log_len
    dom = ArithmeticDomain.of_length(n).with_offset(field.generator())
    coeffs = gctx.synthetic(3 * n, 4242 + log_len)
    want = coeffs.download()
    cw = dom.evaluate(gctx, coeffs, n, 3)
    back = dom.interpolate(gctx, cw, 3)
    assert (back.download() == want).all()
    idx = np.array([0, 1, 12345, n // 2 + 7, n - 1], np.uint64)
    got = np.empty((idx.size, 3), np.uint64)
    gctx._check(gctx.lib.tvm_gather_elements(gctx.handle, cw.ptr, 3, idx.ctypes.data, idx.size, got.ctypes.data), "gather")
    points = np.array([[dom.value(int(i)), 0, 0] for i in idx], np.uint64)
    assert (stark.evaluate_at_points(gctx, coeffs, n, points) == got).all()



@pytest.mark.parametrize("log_n,fk,n_cols,sample_cols", [(21, 1, 40, (0, 17, 39)), (22, 1, 24, (0, 23)), (22, 3, 3, (2,)), (21, 3, 7, (6,)), (23, 1, 6, (5,))])
def test_lde_of_longer_traces(gctx, orc, log_n, fk, n_cols, sample_cols):
    """2^21 and 2^22 rows (BASELINE config 3's height): 2048-point axes, the kernels with two positions per work-item and
    8-row tiles (k_lde_pass2_v3 / k_lde_pass3_v3).  Sampled columns through the oracle at sampled rows."""
    ctx, n = gctx, 1 << log_n
    trace_dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(8 * n).with_offset(field.generator())
    mt = MasterTable.from_device(ctx, ctx.synthetic(n_cols * n * fk, seed=21 + fk), ctx.synthetic(n_cols * H * fk, seed=23 + fk), n_cols, n, H,
                                 trace_dom, ev, ev, fk)
    mt.maybe_low_degree_extend_all_columns()
    rng = np.random.default_rng(log_n)
    rows = np.unique(np.concatenate([[0, 1, 7, 8, 15, 16, len(ev) - 1], rng.integers(0, len(ev), 300)])).astype(np.uint64)
    revealed = mt.reveal_rows(rows)
    trace_host = mt.d_trace.download().reshape((n_cols, n) + ((3,) if fk == 3 else ()))
    rnd_host = mt.d_randomizers.download().reshape((n_cols, H) + ((3,) if fk == 3 else ()))
    for c in sample_cols:
        want = orc.lde_table(trace_host[c:c + 1], rnd_host[c:c + 1], odom(orc, ev), fk)
        assert (revealed[:, c] == want[rows.astype(np.int64), 0]).all(), f"column {c}"
    mt.clear_cache()


@pytest.mark.parametrize("log_n,world,rank,fk,n_cols,sample_cols", [(20, 8, 3, 1, 100, (0, 57, 99)), (20, 2, 1, 3, 5, (4,)),
                                                                      (19, 4, 2, 1, 20, (0, 19)), (19, 1, 0, 1, 12, (11,))])
def test_lde_onto_a_ranks_share_of_the_domain(gctx, orc, log_n, world, rank, fk, n_cols, sample_cols):
    """What a rank of the sharded proof extends onto (triton_vm_amd/sharded.py): the rows i = rank (mod world) of the LDT
    domain -- 8 / world cosets of the trace domain, down to a single one -- at the heights whose pass 3 runs as 8-row tiles
    with two workgroups per CU (1024-point axes: 2^19 and 2^20 rows)."""
    from triton_vm_amd.sharded import local_domain

    ctx, n = gctx, 1 << log_n
    trace_dom = ArithmeticDomain.of_length(n)
    ev = local_domain(ArithmeticDomain.of_length(8 * n).with_offset(field.generator()), rank, world)
    mt = MasterTable.from_device(ctx, ctx.synthetic(n_cols * n * fk, seed=31 + fk), ctx.synthetic(n_cols * H * fk, seed=33 + fk), n_cols, n, H,
                                 trace_dom, ev, ev, fk)
    mt.maybe_low_degree_extend_all_columns()
    rng = np.random.default_rng(log_n + world)
    rows = np.unique(np.concatenate([[0, 1, 7, 8, 15, 16, len(ev) - 1], rng.integers(0, len(ev), 300)])).astype(np.uint64)
    revealed = mt.reveal_rows(rows)
    trace_host = mt.d_trace.download().reshape((n_cols, n) + ((3,) if fk == 3 else ()))
    rnd_host = mt.d_randomizers.download().reshape((n_cols, H) + ((3,) if fk == 3 else ()))
    for c in sample_cols:
        want = orc.lde_table(trace_host[c:c + 1], rnd_host[c:c + 1], odom(orc, ev), fk)
        assert (revealed[:, c] == want[rows.astype(np.int64), 0]).all(), f"column {c}"
    mt.clear_cache()


@pytest.mark.parametrize("log_n,log_ldt_expansion", [(20, 3), (18, 5)])
def test_full_size_air_sampled_rows(gctx, orc, log_n, log_ldt_expansion):
    """all_quotients_combined at BASELINE config 1's size (2^20 rows: quotient = LDT domain 2^23, tables of 23.7 +
    17.1 GiB, i.e. cell offsets far beyond 4 GiB and 32768 AIR workgroups) and on the stride-4 view of a log-blowup-4-
    shaped instance (2^18 rows, LDT domain 2^23, quotient domain 2^21): for sampled quotient-domain rows i -- the first
    and last blocks, the rows whose successor is a wrap row (i >= |Q| - |Q|/N), rows in the upper half of the table --
    rows i and i + |Q|/N of both extended tables are revealed and pushed through the oracle's constraint DAG."""
    from tests.test_kernels_air import check_sampled_quotient_rows

    check_sampled_quotient_rows(gctx, orc, log_n, log_ldt_expansion, H, synthetic=True)


def test_full_size_extend_satisfies_the_transition_constraints(gctx, orc):
    """tvm_extend_aux_table + the degree-lowering fill at 2^20 rows (1024 scan tiles per column).  The table is the valid
    2048-row trace of `program_executing_every_instruction` tiled 512 times: inside a tile the main rows are a valid
    execution, and every transition constraint on an auxiliary column has the form aux' = f(aux, row, row') -- it holds
    for ANY incoming value, so the (reference-pinned) AIR must vanish on sampled row pairs away from the tile seams,
    with running values that were carried across a million rows and a thousand workgroups."""
    from tests import vm_fixture as vf
    from triton_vm_amd import master_table as mtab

    main, aux, ch, _ = vf.valid_tables("every")
    tile = main.shape[1]
    reps = (1 << LOG_N) // tile
    n = tile * reps
    big = np.ascontiguousarray(np.tile(main, (1, reps)))
    d_main = gctx.to_device(big)
    rng = np.random.default_rng(31)
    start = np.zeros((91, n, 3), np.uint64)
    start[90] = orc.random_elements(rng, (n, 3))
    d_aux = gctx.to_device(start)
    mtab.extend(gctx, d_main, d_aux, n, ch)
    got = d_aux.download((91, n, 3))
    assert (got[90] == start[90]).all()
    # the first tile is the valid trace itself: identical to the oracle's extension (the transition-derived columns
    # of the tile's last row excepted: that row has a successor here)
    assert (got[:49, :tile] == aux[:49]).all()
    assert (got[49:90, :tile - 1] == aux[49:90, :tile - 1]).all()
    rows = np.unique(np.concatenate([rng.integers(0, n - 1, 300), [tile + 5, n // 2 + 17, n - tile + 100]]))
    rows = [int(i) for i in rows if i % tile < tile - 2]
    for i in rows:
        v = orc.air_constraint_values(np.ascontiguousarray(big[:, i]), np.ascontiguousarray(big[:, i + 1]),
                                      np.ascontiguousarray(got[:, i]), np.ascontiguousarray(got[:, i + 1]), ch)
        assert not v[81:178].any(), f"consistency constraints, row {i}"
        assert not v[178:581].any(), f"transition constraints, rows {i}, {i + 1}"
