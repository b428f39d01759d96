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


# Sampled columns: the table is extended in launches of 96 base-field columns (csrc/ntt.hip: lde_table's chunks), so there is one
# column inside every launch and one on either side of every launch boundary -- main words 95|96, 191|192, 287|288; the
# auxiliary table's words 95|96 and 191|192 are the XFE columns 31|32 and 63|64 (words 93-95|96-98, 189-191|192-194).
@pytest.mark.parametrize("fk,n_cols,sample_cols", [(1, 379, (0, 50, 95, 96, 150, 191, 192, 250, 287, 288, 378)),
                                                    (3, 91, (0, 31, 32, 50, 63, 64, 90))])
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


# ---- csrc/poly.hip at BASELINE config 1's size: the streaming stages against the oracle on sampled columns / indices ----------
def _sampled_indices(rng, n, k=200):
    return np.unique(np.concatenate([[0, 1, n // 2 - 1, n // 2, n - 1], rng.integers(0, n, k)])).astype(np.int64)


def _gather(ctx, buf, idx, words=3):
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    got = np.empty((idx.size, words), np.uint64)
    ctx._check(ctx.lib.tvm_gather_elements(ctx.handle, buf.ptr, words, idx.ctypes.data, idx.size, got.ctypes.data), "gather")
    return got


@pytest.mark.parametrize("fk,n_cols", [(1, 379), (3, 91)])
def test_full_size_out_of_domain_rows_and_weighted_sums(gctx, orc, fk, n_cols):
    """out_of_domain_row (master_table.rs:348-390) and weighted_sum_of_columns (:512-542) over the whole 2^20-row trace table:
    the out-of-domain values of sampled columns against the oracle's barycentric evaluation of those columns; the weighted
    sum with weights that are zero outside a sampled set of columns against the oracle's sum over that set (every column
    still passes through the kernel's accumulation)."""
    ctx, n = gctx, 1 << LOG_N
    rng = np.random.default_rng(fk + 40)
    trace_dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(8 * n).with_offset(field.generator())
    mt = MasterTable.from_device(ctx, ctx.synthetic(n_cols * n * fk, seed=41 + fk), ctx.synthetic(n_cols * H * fk, seed=43 + fk), n_cols, n, H,
                                 trace_dom, ev, ev, fk)
    shape = lambda k: (n_cols, k) + ((3,) if fk == 3 else ())
    trace_host, rnd_host = mt.d_trace.download(shape(n)), mt.d_randomizers.download(shape(H))
    cols = sorted({0, 1, n_cols // 2, n_cols - 1} | {int(c) for c in rng.integers(0, n_cols, 4)})
    pts = orc.random_elements(rng, (2, 3))
    got = mt.out_of_domain_rows(pts)                                               # [2][n_cols][3]
    for p in range(2):
        want = orc.out_of_domain_row(np.ascontiguousarray(trace_host[cols]), np.ascontiguousarray(rnd_host[cols]), pts[p], fk)
        assert (got[p][cols] == want).all(), f"point {p}"
    w = np.zeros((n_cols, 3), np.uint64)
    w[cols] = orc.random_elements(rng, (len(cols), 3))
    got = mt.weighted_sum_of_columns(w).download((2 * n, 3))
    want = orc.weighted_sum_of_columns(np.ascontiguousarray(trace_host[cols]), np.ascontiguousarray(rnd_host[cols]), w[cols], fk)
    assert (got == want).all()
    for buf in (mt.d_trace, mt.d_randomizers):
        buf.free()


def test_full_size_quotient_segments(gctx, orc):
    """interpolate_quotient_segments + ldt_domain_segment_polynomials + randomize_quotient_segments (stark.rs:1224-1356) on a
    2^23-point quotient codeword: all five randomized segment polynomials coefficient by coefficient against the oracle's
    (its 2^23-point interpolation), and the [L][5] codeword table at sampled rows against Horner evaluations of those
    polynomials at the domain points."""
    from triton_vm_amd import stark

    ctx = gctx
    rng = np.random.default_rng(77)
    L = 8 << LOG_N
    g = field.generator()
    dom = ArithmeticDomain.of_length(L).with_offset(g)
    n_rand = (H + 1) * 5
    d_cw = ctx.synthetic(3 * L, seed=91)
    rnd = orc.random_elements(rng, (n_rand, 3))
    qs = stark.quotient_segments(ctx, d_cw, dom, dom, rnd)
    seg = orc.interpolate_quotient_segments(d_cw.download((L, 3)), odom(orc, dom))   # [4][L/4][3]
    polys = qs.polys.download((5, qs.poly_len, 3))
    # s_4 = the randomizer; s_i = q_i - zeta^i * s_{i+1}(zeta^4 X), zeta = 3 (stark.rs:1338-1352), checked through point values
    idx = _sampled_indices(rng, L, 40)
    rows = np.empty((idx.size, 15), np.uint64)
    ix = idx.astype(np.uint64)
    ctx._check(ctx.lib.tvm_table_reveal_rows(ctx.handle, qs.table, L, ix.ctypes.data, ix.size, rows.ctypes.data), "rows")
    zeta = field.to_mont(3)
    for j, i in enumerate(idx):
        x = dom.value(int(i))
        xp = np.array([x, 0, 0], np.uint64)
        for k in range(5):                                            # the table is the polynomials evaluated on the LDT domain
            assert (rows[j].reshape(5, 3)[k] == orc.poly_eval_xfe(polys[k], xp)).all(), (i, k)
    assert (polys[4][:n_rand] == rnd).all() and not polys[4][n_rand:].any()
    for _ in range(3):                                               # the recursion at random points
        x = orc.random_elements(rng, 3)
        z4x = x.copy()
        for _k in range(4):
            z4x = np.array([field.mont_mul(int(c), zeta) for c in z4x], np.uint64)
        for i in range(4):
            zi = np.array([field.mont_pow(zeta, i), 0, 0], np.uint64)
            want = orc.xfe_sub(orc.poly_eval_xfe(seg[i], x), orc.xfe_mul(zi, orc.poly_eval_xfe(polys[i + 1], z4x)))
            assert (orc.poly_eval_xfe(polys[i], x) == want).all(), i
    qs.free()


def test_full_size_deep_codeword_and_fri_fold(gctx, orc):
    """deep_codeword with its weighted sum (stark.rs:566-625, 1360-1379) over four 2^23-point codewords at sampled indices
    (one extension-field inversion per component and index through the oracle's scalar arithmetic), and the whole first FRI
    fold 2^23 -> 2^22 (fri.rs:349-366) against the oracle."""
    from triton_vm_amd import stark

    ctx = gctx
    rng = np.random.default_rng(78)
    L = 8 << LOG_N
    dom = ArithmeticDomain.of_length(L).with_offset(field.generator())
    bufs = [ctx.synthetic(3 * L, seed=101 + k) for k in range(4)]
    pts, vals, ws = (orc.random_elements(rng, (4, 3)) for _ in range(3))
    deep = stark.deep_codeword(ctx, bufs, dom, pts, vals, ws)
    idx = _sampled_indices(rng, L)
    got = _gather(ctx, deep, idx)
    at = [_gather(ctx, b, idx) for b in bufs]
    for j, i in enumerate(idx):
        x = np.array([dom.value(int(i)), 0, 0], np.uint64)
        acc = np.zeros(3, np.uint64)
        for k in range(4):
            comp = orc.xfe_mul(orc.xfe_sub(at[k][j], vals[k]), orc.xfe_inv(orc.xfe_sub(x, pts[k])))
            acc = orc.xfe_add(acc, orc.xfe_mul(comp, ws[k]))
        assert (got[j] == acc).all(), i
    ch = orc.random_elements(rng, 3)
    folded = stark.split_and_fold(ctx, bufs[0], dom, ch)
    want = orc.fri_split_and_fold(bufs[0].download((L, 3)), odom(orc, dom), ch)
    assert (folded.download((L // 2, 3)) == want).all()
