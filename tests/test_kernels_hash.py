"""csrc/hash.hip against the oracle, through the C ABI: once on the TEST-ONLY CPU fiber
emulation of the same kernel sources (-m "not gpu") and once on the real MI355X (-m gpu)."""
import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, MasterTable, field


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


@pytest.mark.parametrize("n_cols,fk", [(1, 1), (9, 1), (10, 1), (11, 1), (16, 1), (17, 1), (33, 1), (5, 3), (7, 3)])
def test_row_hashes_and_merkle_tree(ctx, orc, n_cols, fk):
    rng = np.random.default_rng(n_cols * 7 + fk)
    n, h, expansion = 8, 3, 4
    shape_t = (n_cols, n) + ((3,) if fk == 3 else ())
    shape_r = (n_cols, h) + ((3,) if fk == 3 else ())
    trace, rnd = orc.random_elements(rng, shape_t), orc.random_elements(rng, shape_r)
    ev = ArithmeticDomain.of_length(n * expansion).with_offset(field.generator())
    mt = MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), ev, ev, fk)
    mt.maybe_low_degree_extend_all_columns()
    table = orc.lde_table(trace, rnd, odom(orc, ev), fk)
    want = orc.hash_rows(table.reshape(len(ev), -1))
    assert (mt.hash_all_ldt_domain_rows() == want).all()
    nodes = mt.merkle_tree()
    assert (nodes == orc.merkle_tree(want)).all()


@pytest.mark.parametrize("n,expansion", [(2, 1), (2, 4), (2, 2), (4, 2), (8, 8)])
def test_row_hashes_of_short_tables(ctx, orc, n, expansion):
    """fewer rows than one wavefront's sixteen (the row-hashing kernel clamps and drops the idle permutations), and a
    full tile plus change"""
    rng = np.random.default_rng(100 * n + expansion)
    n_cols, h = 13, 1
    trace, rnd = orc.random_elements(rng, (n_cols, n)), orc.random_elements(rng, (n_cols, h))
    ev = ArithmeticDomain.of_length(n * expansion).with_offset(field.generator())
    mt = MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), ev, ev, 1)
    mt.maybe_low_degree_extend_all_columns()
    table = orc.lde_table(trace, rnd, odom(orc, ev), 1)
    assert (mt.hash_all_ldt_domain_rows() == orc.hash_rows(table)).all()


def test_ldt_view_is_strided_subset(ctx, orc):
    """ldt domain shorter than the evaluation domain: rows at stride (master_table.rs:792-801)."""
    rng = np.random.default_rng(3)
    n, h = 8, 2
    trace, rnd = orc.random_elements(rng, (4, n)), orc.random_elements(rng, (4, h))
    quot = ArithmeticDomain.of_length(64).with_offset(field.generator())
    ldt = ArithmeticDomain.of_length(16).with_offset(field.generator())
    mt = MasterTable(ctx, trace, rnd, ArithmeticDomain.of_length(n), quot, ldt, 1)
    mt.maybe_low_degree_extend_all_columns()
    table = orc.lde_table(trace, rnd, odom(orc, quot), 1)
    assert (mt.hash_all_ldt_domain_rows() == orc.hash_rows(table[::4])).all()
    assert (mt.reveal_rows([1, 15]) == table[[4, 60]]).all()


# 8 / 9 / 10 / 12: 16-lanes-per-parent levels, two / three / four / six of them to a launch (k_merkle_subtrees; 14 and 17: seven, and two
# launches of them); 17: all three level kernels, -17: the same with several groups of parents per workgroup; "one": TVM_OPTION_MERKLE_SUBTREES
# = 0, a launch per level (k_merkle_level_lanes)
@pytest.mark.parametrize("log_n,subtrees", [(0, 1), (1, 1), (4, 1), (8, 1), (9, 1), (10, 1), (12, 1), (12, 0), (14, 1), (17, 1), (17, 0), (-17, 1)])
def test_merkle_tree_sizes_and_codeword_tree(ctx, orc, log_n, subtrees, request):
    if abs(log_n) > 12 and ctx.kind == "emu":
        pytest.skip("2^14 / 2^17 leaves take minutes on the fiber emulation; the sizes run on the GPU")
    if not subtrees:
        ctx._check(ctx.lib.tvm_ctx_set_option(ctx.handle, 6, 0), "tvm_ctx_set_option")
        request.addfinalizer(lambda: ctx.lib.tvm_ctx_set_option(ctx.handle, 6, 1))
    if log_n < 0:
        # TVM_OPTION_MERKLE_MIN_WORKGROUPS = 3: 1024 groups of 64 parents on the widest level -> 8 per workgroup
        ctx._check(ctx.lib.tvm_ctx_set_option(ctx.handle, 3, 64), "tvm_ctx_set_option")
        request.addfinalizer(lambda: ctx.lib.tvm_ctx_set_option(ctx.handle, 3, 0))
        log_n = -log_n
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    leaves = orc.random_elements(rng, (n, 5))
    d_nodes = ctx.alloc(10 * n)
    d_leaves = ctx.to_device(leaves)  # keep the buffer alive across the call
    ctx._check(ctx.lib.tvm_merkle_tree(ctx.handle, d_leaves.ptr, n, d_nodes.ptr), "merkle")
    assert (d_nodes.download((2 * n, 5)) == orc.merkle_tree(leaves)).all()
    cw = orc.random_elements(rng, (n, 3))
    d_cw = ctx.to_device(cw)
    ctx._check(ctx.lib.tvm_codeword_merkle_tree(ctx.handle, d_cw.ptr, n, d_nodes.ptr), "cw tree")
    assert (d_nodes.download((2 * n, 5)) == orc.merkle_tree(orc.xfe_to_digest(cw))).all()


def test_row_hashing_reproduces_the_reference_program_digest(ctx, orc):
    """The device's row-hashing kernel against a REFERENCE-held value: a table whose 295 columns are constant (the words of
    `program_executing_every_instruction`, zero randomizers) extends to rows that all equal the program's `to_bwords()`, so
    every row digest must be `program.hash()` as snapshotted at /root/reference/triton-vm/src/stark.rs:4828-4838
    (30 absorb blocks in overwrite mode; see tests/test_oracle_pins.py for the oracle-side pin)."""
    import os

    from oracle.vm import isa

    with open(os.path.join(os.path.dirname(__file__), "golden", "program_every_instruction.tasm")) as f:
        words = orc.to_mont(isa.parse(f.read()).to_bwords())
    n, expansion = 4, 4
    trace = np.repeat(words[:, None], n, axis=1)
    trace_dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(n * expansion).with_offset(field.generator())
    mt = MasterTable(ctx, trace, np.zeros((len(words), 2), np.uint64), trace_dom, ev, ev, 1)
    mt.maybe_low_degree_extend_all_columns()
    digests = orc.from_mont(mt.hash_all_ldt_domain_rows())
    want = [16104359835754349618, 14381287807966156775, 14760563195542097310, 2080121037799184588, 13105746022149139394]
    assert all([int(v) for v in row] == want for row in digests)
