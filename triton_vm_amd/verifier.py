"""Host mirror of the verifier's batch work over the revealed rows
(/root/reference/triton-vm/src/stark.rs:1598-1601, 1620-1660, 1678-1755), over the C ABI (csrc/verify.hip).
The decisions (equalities, Merkle inclusion, Fiat-Shamir) stay with the caller, as in the reference's Verifier."""
import ctypes as C

import numpy as np


def _h(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def row_digests(ctx, rows):
    """Tip5::hash_varlen of every revealed row (leaf digests, stark.rs:1598-1601): rows [n][words] -> [n][5]"""
    rows = _h(rows)
    rows = rows.reshape(rows.shape[0], -1)
    out = np.empty((rows.shape[0], 5), np.uint64)
    ctx._check(ctx.lib.tvm_verifier_row_digests(ctx.handle, rows.ctypes.data, rows.shape[0], rows.shape[1], out.ctypes.data),
               "tvm_verifier_row_digests")
    return out


def deep_values(ctx, main_rows, aux_rows, quotient_rows, row_indices, ldt_domain, weights_main_aux, weights_quot, weights_deep,
                ood_points, ood_values):
    """The value the combination codeword must have at each revealed index (stark.rs:1678-1755): [n][3].
    ood_points / ood_values in the order current row, next row, alpha^4, (zeta * alpha)^4."""
    main_rows, aux_rows, quotient_rows = _h(main_rows), _h(aux_rows), _h(quotient_rows)
    idx = _h(row_indices)
    n = idx.size
    if main_rows.size != n * 379 or aux_rows.size != n * 273 or quotient_rows.size != n * 15:
        raise ValueError("revealed rows must be [n][379], [n][91][3] and [n][5][3]")
    wma, wq, wd = _h(weights_main_aux).reshape(470, 3), _h(weights_quot).reshape(5, 3), _h(weights_deep).reshape(4, 3)
    pts, vals = _h(ood_points).reshape(4, 3), _h(ood_values).reshape(4, 3)
    out = np.empty((n, 3), np.uint64)
    ctx._check(ctx.lib.tvm_verifier_deep_values(ctx.handle, main_rows.ctypes.data, aux_rows.ctypes.data, quotient_rows.ctypes.data,
                                                idx.ctypes.data, n, ldt_domain.c(), wma.ctypes.data, wq.ctypes.data, wd.ctypes.data,
                                                pts.ctypes.data, vals.ctypes.data, out.ctypes.data), "tvm_verifier_deep_values")
    return out
