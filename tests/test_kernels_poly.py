"""csrc/poly.hip against the oracle, through the C ABI: once on the TEST-ONLY CPU fiber emulation of
the same kernel sources (-m "not gpu") and once on the real MI355X (-m gpu)."""
import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, MasterTable, field, stark


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


def make_table(ctx, orc, rng, n, n_cols, h, fk):
    shape = lambda k: (n_cols, k) + ((3,) if fk == 3 else ())
    trace, rnd = orc.random_elements(rng, shape(n)), orc.random_elements(rng, shape(h))
    ev = ArithmeticDomain.of_length(8 * n).with_offset(field.generator())
    return MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), ev, ev, fk), trace, rnd


@pytest.mark.parametrize("fk,n_cols", [(1, 7), (3, 4), (1, 20)])
@pytest.mark.parametrize("log_n,h", [(4, 5), (6, 9), (8, 198), (8, 64), (8, 0)])
def test_out_of_domain_rows(ctx, orc, fk, n_cols, log_n, h):
    rng = np.random.default_rng(fk * 100 + n_cols + log_n)
    mt, trace, rnd = make_table(ctx, orc, rng, 1 << log_n, n_cols, h, fk)
    pts = orc.random_elements(rng, (2, 3))
    got = mt.out_of_domain_rows(pts)
    for p in range(2):
        assert (got[p] == orc.out_of_domain_row(trace, rnd, pts[p], fk)).all()


@pytest.mark.parametrize("fk,n_cols", [(1, 7), (3, 4)])
@pytest.mark.parametrize("log_n,h", [(1, 1), (4, 5), (6, 9)])
def test_weighted_sum_of_columns(ctx, orc, fk, n_cols, log_n, h):
    rng = np.random.default_rng(fk * 10 + n_cols + log_n)
    n = 1 << log_n
    mt, trace, rnd = make_table(ctx, orc, rng, n, n_cols, h, fk)
    w = orc.random_elements(rng, (n_cols, 3))
    got = mt.weighted_sum_of_columns(w).download((2 * n, 3))
    assert (got == orc.weighted_sum_of_columns(trace, rnd, w, fk)).all()


@pytest.mark.parametrize("n", [1, 5, 64, 1000, 5000])
def test_evaluate_at_points(ctx, orc, n):
    rng = np.random.default_rng(n)
    co = orc.random_elements(rng, (n, 3))
    pts = orc.random_elements(rng, (3, 3))
    d = ctx.to_device(co)
    got = stark.evaluate_at_points(ctx, d, n, pts)
    for p in range(3):
        assert (got[p] == orc.poly_eval_xfe(co, pts[p])).all()


@pytest.mark.parametrize("n,stride,n_polys", [(1, 1, 1), (5, 5, 5), (64, 70, 3), (1000, 1000, 5)])
def test_evaluate_polys_at_points(ctx, orc, n, stride, n_polys):
    """tvm_evaluate_polys_at_points: several polynomials (stride XFE apart) at the same points in one round trip -- the five
    quotient-segment polynomials at the two out-of-domain points (stark.rs:474-495)"""
    rng = np.random.default_rng(n + stride)
    co = orc.random_elements(rng, (n_polys * stride, 3))
    pts = orc.random_elements(rng, (2, 3))
    d = ctx.to_device(co)
    out = np.full((n_polys * 2 + 1, 3), 77, np.uint64)
    ctx._check(ctx.lib.tvm_evaluate_polys_at_points(ctx.handle, d.ptr, n, stride, n_polys, pts.ctypes.data, 2, out.ctypes.data),
               "tvm_evaluate_polys_at_points")
    assert (out[-1] == 77).all()
    for k in range(n_polys):
        for j in range(2):
            assert (out[2 * k + j] == orc.poly_eval_xfe(co[k * stride:k * stride + n], pts[j])).all()


@pytest.mark.parametrize("log_q,log_ldt,n_rand", [(4, 4, 3), (5, 5, 8), (5, 7, 5), (6, 6, 16), (9, 11, 7), (8, 9, 100), (7, 5, 32)])
def test_quotient_segments(ctx, orc, log_q, log_ldt, n_rand):
    rng = np.random.default_rng(log_q * 7 + log_ldt + n_rand)
    g = field.generator()
    qd = ArithmeticDomain.of_length(1 << log_q).with_offset(g)
    ldt = ArithmeticDomain.of_length(1 << log_ldt).with_offset(g)
    cw = orc.random_elements(rng, (len(qd), 3))
    rnd = orc.random_elements(rng, (n_rand, 3))
    d_cw = ctx.to_device(cw)
    qs = stark.quotient_segments(ctx, d_cw, qd, ldt, rnd)
    seg = orc.interpolate_quotient_segments(cw, odom(orc, qd))
    want_polys, want_cws = orc.randomize_quotient_segments(seg, rnd, odom(orc, ldt), qs.poly_len)
    assert (qs.polys.download((5, qs.poly_len, 3)) == want_polys).all()
    assert (qs.codewords() == want_cws).all()
    digests = orc.hash_rows(want_cws.reshape(len(ldt), 15))
    assert (qs.merkle_tree() == orc.merkle_tree(digests)).all()
    w = orc.random_elements(rng, (5, 3))
    got = qs.linear_combination(w).download((len(ldt), 3))
    want = np.zeros((len(ldt), 3), np.uint64)
    for i in range(len(ldt)):
        acc = np.zeros(3, np.uint64)
        for k in range(5):
            acc = orc.xfe_add(acc, orc.xfe_mul(want_cws[i, k], w[k]))
        want[i] = acc
    assert (got == want).all()


@pytest.mark.parametrize("log_n,expansion,view_stride,n_cols", [(13, 1, 1, 5), (10, 4, 1, 5), (8, 16, 1, 5), (10, 8, 2, 5), (6, 4, 1, 5),
                                                                 (10, 4, 1, 4)])
def test_table_linear_combination_in_tiles(ctx, orc, log_n, expansion, view_stride, n_cols):
    """tvm_table_linear_combination over tables stored coset-major in the order of the last LDE pass: the tiled kernel (16 rows x
    128 (block, coset) pairs per workgroup, results handed through shared memory into runs of consecutive domain rows) for a
    single coset (a rank's table at eight ranks), for 4 and 16 cosets, for a stride view, for a column count other than the
    quotient-segment table's five (the form with loops as they come) -- and the plain kernel for a table too small for a tile --
    against the weighted sum of the exported rows"""
    import ctypes as C

    from triton_vm_amd.master_table import MasterTable

    rng = np.random.default_rng(log_n * 31 + expansion)
    n, h = 1 << log_n, 3
    trace_dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(expansion * n).with_offset(field.generator())
    mt = MasterTable(ctx, orc.random_elements(rng, (n_cols, n, 3)), orc.random_elements(rng, (n_cols, h, 3)), trace_dom, ev, ev, 3)
    mt.maybe_low_degree_extend_all_columns()
    rows = mt.low_degree_extended_table()                      # [L, 5, 3] in domain order
    w = orc.random_elements(rng, (n_cols, 3))
    n_out = len(ev) // view_stride
    out = ctx.alloc(3 * n_out)
    ctx._check(ctx.lib.tvm_table_linear_combination(ctx.handle, C.c_void_p(mt._table), n_out, w.ctypes.data, out.ptr), "lincomb")
    got = out.download((n_out, 3))
    for i in range(n_out):
        acc = np.zeros(3, np.uint64)
        for c in range(n_cols):
            acc = orc.xfe_add(acc, orc.xfe_mul(rows[i * view_stride, c], w[c]))
        assert (got[i] == acc).all(), i
    mt.clear_cache()


@pytest.mark.parametrize("log_n,k", [(1, 2), (3, 1), (6, 4), (8, 2)])
def test_deep_codeword(ctx, orc, log_n, k):
    rng = np.random.default_rng(log_n + k)
    dom = ArithmeticDomain.of_length(1 << log_n).with_offset(field.generator())
    cws = [orc.random_elements(rng, (len(dom), 3)) for _ in range(k)]
    pts, vals, ws = (orc.random_elements(rng, (k, 3)) for _ in range(3))
    bufs = [ctx.to_device(c) for c in cws]
    got = stark.deep_codeword(ctx, bufs, dom, pts, vals, ws).download((len(dom), 3))
    want = np.zeros((len(dom), 3), np.uint64)
    for j in range(k):
        comp = orc.deep_codeword(cws[j], odom(orc, dom), pts[j], vals[j])
        for i in range(len(dom)):
            want[i] = orc.xfe_add(want[i], orc.xfe_mul(comp[i], ws[j]))
    assert (got == want).all()


@pytest.mark.parametrize("log_n", [1, 2, 5, 9])
def test_fri_split_and_fold_and_round_tree(ctx, orc, log_n):
    rng = np.random.default_rng(log_n)
    dom = ArithmeticDomain.of_length(1 << log_n).with_offset(field.generator())
    cw = orc.random_elements(rng, (len(dom), 3))
    ch = orc.random_elements(rng, 3)
    d = ctx.to_device(cw)
    folded = stark.split_and_fold(ctx, d, dom, ch)
    want = orc.fri_split_and_fold(cw, odom(orc, dom), ch)
    assert (folded.download((len(dom) // 2, 3)) == want).all()
    nodes = stark.merkle_tree_from_codeword(ctx, folded, len(dom) // 2).download((len(dom), 5))
    assert (nodes == orc.merkle_tree(orc.xfe_to_digest(want))).all()


def test_gather_elements_batch(ctx):
    """tvm_gather_elements_batch == one tvm_gather_elements per job (jobs of different element widths, an empty one, repeated
    indices)"""
    import ctypes as C

    rng = np.random.default_rng(5)
    shapes = [(1000, 3), (77, 5), (64, 15), (10, 1)]
    hosts = [rng.integers(0, 1 << 63, size=s, dtype=np.uint64) for s in shapes]
    bufs = [ctx.to_device(h) for h in hosts]
    idxs = [rng.integers(0, s[0], size=k).astype(np.uint64) for s, k in zip(shapes, (173, 9, 0, 4))]
    outs = [np.zeros((i.size, s[1]), np.uint64) for i, s in zip(idxs, shapes)]
    n_jobs = len(shapes)
    src = (C.c_void_p * n_jobs)(*[b.ptr for b in bufs])
    words = (C.c_uint32 * n_jobs)(*[s[1] for s in shapes])
    hidx = (C.c_void_p * n_jobs)(*[i.ctypes.data if i.size else None for i in idxs])
    ns = (C.c_uint64 * n_jobs)(*[i.size for i in idxs])
    hout = (C.c_void_p * n_jobs)(*[o.ctypes.data if o.size else None for o in outs])
    ctx._check(ctx.lib.tvm_gather_elements_batch(ctx.handle, n_jobs, src, words, hidx, ns, hout), "gather batch")
    for h, i, o in zip(hosts, idxs, outs):
        assert (o == h[i.astype(np.int64)]).all()
    assert ctx.lib.tvm_gather_elements_batch(ctx.handle, 0, None, None, None, None, None) == 0


@pytest.mark.parametrize("log_n,n_rounds", [(6, 3), (9, 5), (4, 0)])
def test_fri_commit_phase_equals_the_round_by_round_path(ctx, orc, log_n, n_rounds):
    """tvm_fri_commit_phase (trees, transcript and folds without the host in the loop: the sponge runs on the device) ==
    tvm_codeword_merkle_tree + ProofStream.enqueue / sample_scalars on the host + tvm_fri_split_and_fold, round by round"""
    import ctypes as C

    from triton_vm_amd.arithmetic_domain import ArithmeticDomain
    from triton_vm_amd.proof_stream import ProofStream

    rng = np.random.default_rng(11)
    n = 1 << log_n
    dom = ArithmeticDomain.of_length(n).with_offset(field.to_mont(7))
    cw = orc.random_elements(rng, (n, 3))
    d_cw = ctx.to_device(cw)
    ps = ProofStream(ctx.lib)
    ps.enqueue("log2 padded height", np.array([field.to_mont(5)], np.uint64))     # some history in the sponge
    state0 = ps.state.copy()
    # the device path
    cws = [ctx.alloc(max(n >> (r + 1), 1) * 3) for r in range(n_rounds)]
    nodes = [ctx.alloc(10 * (n >> r)) for r in range(n_rounds + 1)]
    roots = np.zeros((n_rounds + 1, 5), np.uint64)
    challenges = np.zeros((max(n_rounds, 1), 3), np.uint64)
    p_cws = (C.c_void_p * max(n_rounds, 1))(*[b.ptr for b in cws])
    p_nodes = (C.c_void_p * (n_rounds + 1))(*[b.ptr for b in nodes])
    ctx._check(ctx.lib.tvm_fri_commit_phase(ctx.handle, d_cw.ptr, dom.c(), n_rounds, state0.ctypes.data, p_cws, p_nodes,
                                            roots.ctypes.data, challenges.ctypes.data), "commit phase")
    # round by round through the host
    cur, d = d_cw, dom
    for r in range(n_rounds + 1):
        tree = stark.merkle_tree_from_codeword(ctx, cur, d.length)
        assert (tree.download((2 * d.length, 5)) [1:] == nodes[r].download((2 * d.length, 5))[1:]).all()
        root = tree.download((2 * d.length, 5))[1]
        assert (root == roots[r]).all()
        ps.enqueue(f"fri root {r}", root)
        if r == n_rounds:
            break
        ch = ps.sample_scalars(1)[0]
        assert (np.asarray(ch, np.uint64) == challenges[r]).all()
        nxt = stark.split_and_fold(ctx, cur, d, ch)
        assert (nxt.download((d.length // 2, 3)) == cws[r].download((d.length // 2, 3))).all()
        cur, d = nxt, d.pow(2)
