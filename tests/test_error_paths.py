"""Errors cross the boundary as status codes, never as aborts (`Result<_, ProvingError>` -> int32 status; a
refused call leaves the context usable).  Malformed calls of the hot path's entry points -- null pointers, lengths that
are no power of two, domains that do not match the data, more randomizers than rows, too many DEEP components --
must return TVM_ERR_INVALID_ARGUMENT / TVM_ERR_UNSUPPORTED with a message, and a well-formed call afterwards must work."""
import ctypes as C

import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, field
from triton_vm_amd.capi import TritonHipError

INVALID, UNSUPPORTED = 1, 4


def refused(ctx, status, what):
    assert status in (INVALID, UNSUPPORTED), (what, status)
    assert ctx.lib.tvm_last_error(ctx.handle), what          # a message is kept for the host to report


def test_malformed_calls_are_refused_and_the_context_survives(ctx, orc):
    lib, h = ctx.lib, ctx.handle
    n = 16
    dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(4 * n).with_offset(field.generator())
    rng = np.random.default_rng(5)
    trace = orc.random_elements(rng, (3, n))
    rnd = orc.random_elements(rng, (3, 2))
    d_trace, d_rnd = ctx.to_device(trace), ctx.to_device(rnd)
    t = C.c_void_p()

    # tvm_lde_table: null trace, zero columns, a trace domain of another length, more randomizers than rows, bad field kind
    refused(ctx, lib.tvm_lde_table(h, 1, None, n, 3, d_rnd.ptr, 2, dom.c(), ev.c(), C.byref(t)), "null trace")
    refused(ctx, lib.tvm_lde_table(h, 1, d_trace.ptr, n, 0, d_rnd.ptr, 2, dom.c(), ev.c(), C.byref(t)), "no columns")
    refused(ctx, lib.tvm_lde_table(h, 1, d_trace.ptr, n, 3, d_rnd.ptr, 2, ArithmeticDomain.of_length(2 * n).c(), ev.c(), C.byref(t)), "domain length")
    refused(ctx, lib.tvm_lde_table(h, 1, d_trace.ptr, n, 3, d_rnd.ptr, n + 1, dom.c(), ev.c(), C.byref(t)), "h > n")
    refused(ctx, lib.tvm_lde_table(h, 2, d_trace.ptr, n, 3, d_rnd.ptr, 2, dom.c(), ev.c(), C.byref(t)), "field kind 2")
    refused(ctx, lib.tvm_lde_table(h, 1, d_trace.ptr, n, 3, None, 2, dom.c(), ev.c(), C.byref(t)), "randomizers missing")
    assert not t.value
    # transforms of a length that is no power of two; a domain whose generator is zero
    d_buf = ctx.alloc(3 * 64)
    refused(ctx, lib.tvm_ntt(h, 1, d_buf.ptr, 12, dom.generator), "ntt of 12 points")
    bad = dom.c()
    bad.length = 12
    refused(ctx, lib.tvm_evaluate(h, 1, d_buf.ptr, 4, bad, d_buf.ptr), "evaluate on a domain of 12 points")
    # Merkle trees of 0 and of 3 leaves
    d_nodes = ctx.alloc(10 * 64)
    refused(ctx, lib.tvm_merkle_tree(h, d_buf.ptr, 0, d_nodes.ptr), "tree of no leaves")
    refused(ctx, lib.tvm_merkle_tree(h, d_buf.ptr, 3, d_nodes.ptr), "tree of three leaves")
    refused(ctx, lib.tvm_merkle_tree(h, None, 8, d_nodes.ptr), "tree of null leaves")
    # DEEP with five components, FRI fold of a null codeword
    ptrs = (C.c_void_p * 5)(*[d_buf.ptr] * 5)
    z = np.zeros(15, np.uint64)
    refused(ctx, lib.tvm_deep_codeword(h, 5, ptrs, dom.c(), z.ctypes.data, z.ctypes.data, z.ctypes.data, d_buf.ptr), "five DEEP components")
    refused(ctx, lib.tvm_fri_split_and_fold(h, None, dom.c(), z.ctypes.data, d_buf.ptr), "fold of null")
    # hashing rows of a null table, quotients without tables
    refused(ctx, lib.tvm_hash_rows(h, None, 64, d_nodes.ptr), "rows of no table")
    refused(ctx, lib.tvm_all_quotients_combined(h, None, None, dom.c(), ev.c(), z.ctypes.data, z.ctypes.data, d_buf.ptr), "quotients of no tables")

    # ... and the context still works: the well-formed extension matches the oracle
    ctx._check(lib.tvm_lde_table(h, 1, d_trace.ptr, n, 3, d_rnd.ptr, 2, dom.c(), ev.c(), C.byref(t)), "tvm_lde_table")
    assert t.value and lib.tvm_table_num_rows(t) == 4 * n and lib.tvm_table_num_columns(t) == 3
    d_out = ctx.alloc(4 * n * 3)                       # the export writes DEVICE memory
    ctx._check(lib.tvm_table_export_row_major(h, t, d_out.ptr), "export")
    out = d_out.download((4 * n, 3))
    want = orc.lde_table(trace, rnd, orc.Domain(ev.offset, ev.generator, ev.length), 1)
    assert (out == want).all()
    lib.tvm_table_free(h, t)


def test_the_python_wrapper_raises_with_the_message(ctx):
    with pytest.raises(TritonHipError, match="invalid argument"):
        ctx._check(ctx.lib.tvm_ntt(ctx.handle, 1, None, 8, field.ONE), "tvm_ntt")


def test_the_split_extension_refuses_foreign_handles_other_domains_and_incomplete_tables(ctx, orc):
    """tvm_lde_table_begin / _add_columns / _end (the column split, include/triton_hip.h): add_columns and end take only a table that
    begin made and end has not closed, only with begin's domains and randomizer count, only columns inside the table (no wrap-around
    in first + count), and end refuses a table with a column that was never written (round-5 advice)."""
    lib, h = ctx.lib, ctx.handle
    n, n_cols, hr = 64, 5, 3
    dom = ArithmeticDomain.of_length(n)
    ev = ArithmeticDomain.of_length(4 * n).with_offset(field.generator())
    rng = np.random.default_rng(9)
    trace, rnd = orc.random_elements(rng, (n_cols, n)), orc.random_elements(rng, (n_cols, hr))
    d_trace, d_rnd = ctx.to_device(trace), ctx.to_device(rnd)
    coeffs = ctx.alloc(n_cols * n)
    ctx._check(lib.tvm_lde_column_coefficients(h, 1, d_trace.ptr, n, n_cols, dom.c(), 0, n_cols, coeffs.ptr), "coefficients")
    refused(ctx, lib.tvm_lde_column_coefficients(h, 1, d_trace.ptr, n, n_cols, dom.c(), 2, 2 ** 64 - 1, coeffs.ptr), "first + count wraps")

    whole = C.c_void_p()
    ctx._check(lib.tvm_lde_table(h, 1, d_trace.ptr, n, n_cols, d_rnd.ptr, hr, dom.c(), ev.c(), C.byref(whole)), "tvm_lde_table")
    refused(ctx, lib.tvm_lde_table_add_columns(h, whole, coeffs.ptr, 0, n_cols, d_rnd.ptr, hr, dom.c(), ev.c()), "add_columns on a finished table")
    refused(ctx, lib.tvm_lde_table_end(h, whole), "end on a table that begin did not make")

    t = C.c_void_p()
    ctx._check(lib.tvm_lde_table_begin(h, 1, n, n_cols, hr, dom.c(), ev.c(), C.byref(t)), "begin")
    other_ev = ArithmeticDomain.of_length(4 * n).with_offset(field.mont_mul(field.generator(), field.generator()))
    refused(ctx, lib.tvm_lde_table_add_columns(h, t, coeffs.ptr, 0, n_cols, d_rnd.ptr, hr, dom.c(), other_ev.c()), "another evaluation domain")
    refused(ctx, lib.tvm_lde_table_add_columns(h, t, coeffs.ptr, 0, n_cols, d_rnd.ptr, hr + 1, dom.c(), ev.c()), "another randomizer count")
    refused(ctx, lib.tvm_lde_table_add_columns(h, t, coeffs.ptr, 3, 2 ** 64 - 2, d_rnd.ptr, hr, dom.c(), ev.c()), "first + count wraps")
    refused(ctx, lib.tvm_lde_table_add_columns(h, t, coeffs.ptr, 4, 2, d_rnd.ptr, hr, dom.c(), ev.c()), "columns beyond the table")
    ctx._check(lib.tvm_lde_table_add_columns(h, t, coeffs.ptr, 0, 3, d_rnd.ptr, hr, dom.c(), ev.c()), "columns 0..2")
    refused(ctx, lib.tvm_lde_table_end(h, t), "end with columns 3, 4 never written")
    ctx._check(lib.tvm_lde_table_add_columns(h, t, coeffs.ptr + 8 * 3 * n, 3, 2, d_rnd.ptr, hr, dom.c(), ev.c()), "columns 3, 4")
    ctx._check(lib.tvm_lde_table_end(h, t), "end")
    refused(ctx, lib.tvm_lde_table_end(h, t), "end twice")
    # ... and the assembled table is the whole one
    out_a, out_b = ctx.alloc(4 * n * n_cols), ctx.alloc(4 * n * n_cols)
    ctx._check(lib.tvm_table_export_row_major(h, t, out_a.ptr), "export")
    ctx._check(lib.tvm_table_export_row_major(h, whole, out_b.ptr), "export")
    assert (out_a.download() == out_b.download()).all()
    lib.tvm_table_free(h, t)
    lib.tvm_table_free(h, whole)
