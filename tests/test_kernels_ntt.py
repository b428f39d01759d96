"""csrc/ntt.hip against the oracle, through the C ABI: once on the TEST-ONLY CPU fiber emulation of
the same kernel sources (-m "not gpu") and once on the real MI355X (-m gpu)."""
import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, MasterTable, field


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 6, 9])
@pytest.mark.parametrize("fk", [1, 3])
def test_ntt_intt_match_oracle(ctx, orc, log_n, fk):
    rng = np.random.default_rng(log_n * 10 + fk)
    n = 1 << log_n
    a = orc.random_elements(rng, n * fk)
    gen = field.primitive_root_of_unity(n)
    d = ctx.to_device(a)
    ctx._check(ctx.lib.tvm_ntt(ctx.handle, fk, d.ptr, n, gen), "ntt")
    assert (d.download() == orc.ntt(a, fk)).all()
    ctx._check(ctx.lib.tvm_intt(ctx.handle, fk, d.ptr, n, gen), "intt")
    assert (d.download() == a).all()


@pytest.mark.parametrize("log_len,n_coeffs", [(3, 8), (5, 20), (6, 9), (4, 40), (7, 1), (2, 3), (0, 5), (5, 0)])
@pytest.mark.parametrize("fk", [1, 3])
def test_evaluate_interpolate_match_oracle(ctx, orc, log_len, n_coeffs, fk):
    rng = np.random.default_rng(log_len * 100 + n_coeffs + fk)
    dom = ArithmeticDomain.of_length(1 << log_len).with_offset(field.generator())
    co = orc.random_elements(rng, max(n_coeffs, 1) * fk)[: n_coeffs * fk]
    got = dom.evaluate(ctx, ctx.to_device(co) if n_coeffs else None, n_coeffs, fk).download()
    want = orc.coset_evaluate(co, odom(orc, dom), fk)
    assert (got == want).all()
    back = dom.interpolate(ctx, ctx.to_device(want), fk).download()
    assert (back == orc.coset_interpolate(want, odom(orc, dom), fk)).all()


@pytest.mark.parametrize("log_n,expansion,n_cols,h", [(4, 8, 3, 5), (5, 4, 18, 7), (6, 8, 33, 20), (3, 2, 1, 8), (1, 4, 2, 1),
                                                       (12, 8, 2, 70), (13, 4, 1, 9), (12, 2, 1, 4096),
                                                       (4, 1, 3, 5), (12, 1, 2, 7),   # expansion 1: a rank's share at 8 GPUs
                                                       # 2^13 / 2^14 / 2^15 rows: the kernels with two positions per work-item and
                                                       # 8-row tiles (the shape of 2^21 / 2^22-row traces)
                                                       (13, 8, 3, 198), (14, 8, 2, 198), (14, 4, 17, 5), (15, 2, 2, 9000), (14, 1, 2, 3),
                                                       # ... and four positions per work-item, 4-row tiles (2^23 / 2^24 rows)
                                                       (15, 8, 2, 20), (16, 4, 2, 198), (16, 1, 1, 3), (17, 2, 1, 70)])
@pytest.mark.parametrize("fk", [1, 3])
def test_lde_table_matches_oracle(ctx, orc, log_n, expansion, n_cols, h, fk):
    rng = np.random.default_rng(log_n + 31 * n_cols + fk)
    n = 1 << log_n
    shape_t = (n_cols, n) + ((3,) if fk == 3 else ())
    shape_r = (n_cols, h) + ((3,) if fk == 3 else ())
    trace, rnd = orc.random_elements(rng, shape_t), orc.random_elements(rng, shape_r)
    trace_dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(n * expansion).with_offset(field.generator())
    mt = MasterTable(ctx, trace, rnd, trace_dom, ev, ev, fk)
    mt.maybe_low_degree_extend_all_columns()
    got = mt.low_degree_extended_table()
    want = orc.lde_table(trace, rnd, odom(orc, ev), fk)
    assert got.shape == want.shape
    assert (got == want).all()
    # reveal_rows is a gather of the same table
    idx = [0, 3, len(ev) - 1]
    assert (mt.reveal_rows(idx) == want[idx]).all()


@pytest.mark.parametrize("log_n,expansion,n_cols,h,fk,blocks,chunks", [
    (3, 8, 5, 4, 1, 2, 1), (5, 1, 7, 3, 3, 8, 2), (6, 4, 33, 20, 1, 4, 3), (12, 2, 3, 70, 3, 2, 2),
    (13, 1, 3, 198, 1, 8, 1), (14, 2, 2, 198, 3, 2, 1),   # two positions per work-item (the shape of 2^21 / 2^22-row traces)
    (15, 1, 2, 20, 1, 2, 1),                               # four positions per work-item (2^23 / 2^24 rows)
    (19, 1, 2, 198, 1, 2, 1),                              # one row per wavefront (k_lde_pass2_fused: 2^19 / 2^20 rows)
    pytest.param(20, 1, 9, 198, 1, 8, 2, marks=pytest.mark.gpu), pytest.param(20, 8, 3, 198, 3, 2, 1, marks=pytest.mark.gpu),
    pytest.param(22, 1, 3, 198, 1, 8, 1, marks=pytest.mark.gpu)])
def test_lde_split_at_the_coefficients_gives_the_same_table(ctx, orc, log_n, expansion, n_cols, h, fk, blocks, chunks):
    """tvm_lde_column_coefficients + tvm_lde_table_begin / _add_columns / _end (the column sharding of SURVEY 8(e): blocks of
    columns interpolated separately, the table assembled chunk by chunk from the exchanged coefficient form) == tvm_lde_table, word
    for word, on every pass-2 kernel: generic, two and four positions per work-item, one row per wavefront (2^20 rows, GPU)."""
    rng = np.random.default_rng(log_n + 7 * n_cols + fk)
    n = 1 << log_n
    shape = lambda k: (n_cols, k) + ((3,) if fk == 3 else ())
    trace, rnd = orc.random_elements(rng, shape(n)), orc.random_elements(rng, shape(h))
    ev = ArithmeticDomain.of_length(n * expansion).with_offset(field.generator())
    mt = MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), ev, ev, fk)
    mt.maybe_low_degree_extend_all_columns()
    rows = np.unique(np.concatenate([[0, 1, len(ev) - 1], rng.integers(0, len(ev), 500)])).astype(np.uint64)
    want = mt.reveal_rows(rows) if log_n > 15 else mt.low_degree_extended_table()
    mt.low_degree_extend_by_column_blocks(blocks, chunks)
    got = mt.reveal_rows(rows) if log_n > 15 else mt.low_degree_extended_table()
    assert (got == want).all()
    if log_n <= 12:
        assert (got == orc.lde_table(trace, rnd, odom(orc, ev), fk)).all()
    mt.clear_cache()


@pytest.mark.parametrize("log_n,fk,n_cols", [(20, 1, 2), (19, 3, 1), (21, 1, 1), (22, 1, 1),
                                             (16, 1, 3), (17, 3, 1), (18, 1, 2)])   # round 6: 256- and 512-point rows (k_lde_pass2_fused<8 | 9> ...)
def test_lde_with_1024_point_axes(ctx, orc, log_n, fk, n_cols):
    """The production kernels of tvm_lde_table (one transform row per wavefront, csrc/ntt.hip: k_lde_pass2_fused /
    k_lde_pass3_rows, taken for 1024-point axes: traces of 2^19 and 2^20 rows -- BASELINE config 1's height; since round 4 also for
    2048-point axes: 2^21 rows (pass 2) and, on the GPU, 2^22 rows -- BASELINE config 2's height -- in all passes that have them) on a few
    columns at full height, sampled rows against the oracle's extension of the same columns.  (On the emulation too: its
    wavefront-level synchronisation models the wavefront-private LDS exchange of these kernels.)"""
    rng = np.random.default_rng(log_n + fk)
    n, h = 1 << log_n, 198
    shape = lambda k: (n_cols, k) + ((3,) if fk == 3 else ())
    trace, rnd = orc.random_elements(rng, shape(n)), orc.random_elements(rng, shape(h))
    ev = ArithmeticDomain.of_length(8 * n).with_offset(field.generator())
    mt = MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), ev, ev, fk)
    mt.maybe_low_degree_extend_all_columns()
    rows = np.unique(np.concatenate([[0, 1, 7, 8, 15, 16, 8 * n - 1], rng.integers(0, 8 * n, 3000)])).astype(np.uint64)
    got = mt.reveal_rows(rows)
    want = orc.lde_table(trace, rnd, orc.Domain(ev.offset, ev.generator, ev.length), fk)
    assert (got == want[rows.astype(np.int64)]).all()
    mt.clear_cache()


@pytest.mark.parametrize("log_n,h", [(19, 1), (19, 512), (19, 513), (21, 1024), (21, 1025)])
def test_lde_with_1024_point_axes_at_the_randomizer_bound(ctx, orc, log_n, h):
    """k_lde_pass2_fused is taken while the randomizer polynomial has no more coefficients than a transform row has points
    (h <= n1: its term then touches the first coefficient group only; csrc/ntt.hip's dispatcher); one more and the tile kernel takes
    over.  2^19 rows = 512 x 1024 (one wavefront per row) and 2^21 rows = 1024 x 2048 (two): h = 1, h = n1 and h = n1 + 1 against the
    oracle's extension, sampled rows."""
    fk, n_cols = 1, 1
    rng = np.random.default_rng(1000 + h)
    n = 1 << log_n
    trace, rnd = orc.random_elements(rng, (n_cols, n)), orc.random_elements(rng, (n_cols, h))
    ev = ArithmeticDomain.of_length(8 * n).with_offset(field.generator())
    mt = MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), ev, ev, fk)
    mt.maybe_low_degree_extend_all_columns()
    rows = np.unique(np.concatenate([[0, 1, 7, 8, 8 * n - 1], rng.integers(0, 8 * n, 2000)])).astype(np.uint64)
    got = mt.reveal_rows(rows)
    want = orc.lde_table(trace, rnd, orc.Domain(ev.offset, ev.generator, ev.length), fk)
    assert (got == want[rows.astype(np.int64)]).all()
    mt.clear_cache()


@pytest.mark.parametrize("log_n,n_cols", [(6, 7), (12, 7), (16, 2), (17, 2), (19, 3), (21, 2)])
def test_lde_tuning_options_never_change_the_table(ctx, orc, log_n, n_cols, request):
    """include/triton_hip.h: TVM_OPTION_LDE_CHUNK_COLUMNS and TVM_OPTION_LDE_PASS2_TILES "change launch shapes, never results" --
    chunks of 1 / 3 / 4096 columns (ragged last chunks; bench.py sets 32 for the lockstep ranks at 2^22 rows) and the tile kernel
    instead of the row kernels on 2048-point axes (2^21 rows), against the default extension of the same table."""
    rng = np.random.default_rng(77 + log_n)
    n, h = 1 << log_n, min(198, 1 << log_n)
    trace, rnd = orc.random_elements(rng, (n_cols, n)), orc.random_elements(rng, (n_cols, h))
    ev = ArithmeticDomain.of_length(4 * n).with_offset(field.generator())
    rows = np.unique(np.concatenate([[0, 1, 4 * n - 1], rng.integers(0, 4 * n, 600)])).astype(np.uint64)
    set_option = lambda option, value: ctx._check(ctx.lib.tvm_ctx_set_option(ctx.handle, option, value), "tvm_ctx_set_option")
    request.addfinalizer(lambda: (ctx.lib.tvm_ctx_set_option(ctx.handle, 2, 0), ctx.lib.tvm_ctx_set_option(ctx.handle, 4, 0)))

    def table():
        mt = MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), ev, ev, 1)
        mt.maybe_low_degree_extend_all_columns()
        got = mt.reveal_rows(rows)
        mt.clear_cache()
        return got

    want = table()
    if log_n <= 12:
        assert (want == orc.lde_table(trace, rnd, orc.Domain(ev.offset, ev.generator, ev.length), 1)[rows.astype(np.int64)]).all()
    for chunk, tiles in ((1, 0), (3, 0), (4096, 0), (0, 1), (3, 1)):
        set_option(2, chunk)
        set_option(4, tiles)
        assert (table() == want).all(), (chunk, tiles)
