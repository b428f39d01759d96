"""The degree-lowering fill on the device: host-side mirror of the reference's generated
`DegreeLoweringTable::fill_derived_main_columns / fill_derived_aux_columns`
(/root/reference/triton-constraint-builder/src/substitutions.rs:128-205), which
`MasterMainTable::pad` and `MasterMainTable::extend` call as their last step
(/root/reference/triton-vm/src/table/master_table.rs:980-982, 1066-1072).

The tables are the column-major traces `MasterTable` uploads: main `[379][n]` words, aux `[>= 90][n][3]` words;
the derived columns (main 149..378, aux 49..89) are written in place.  Same names and argument meaning as the
reference; the work is in libtriton_hip.so (csrc/fill.h, csrc/fill_gen.hip) -- there is no host fallback.
"""
import ctypes as C

import numpy as np

NUM_MAIN_COLUMNS, NUM_AUX_COLUMNS = 379, 90          # aux: without the randomizer column
FIRST_DERIVED_MAIN, FIRST_DERIVED_AUX = 149, 49
NUM_CHALLENGES = 63


def fill_derived_main_columns(ctx, d_main_trace, n_rows):
    """d_main_trace: DeviceBuffer of [379][n_rows] words (columns 0..148 filled); derives columns 149..378."""
    if d_main_trace.n_words < NUM_MAIN_COLUMNS * n_rows:
        raise ValueError("the main trace needs 379 columns")
    ctx._check(ctx.lib.tvm_fill_derived_main_columns(ctx.handle, d_main_trace.ptr, n_rows), "tvm_fill_derived_main_columns")


def fill_derived_aux_columns(ctx, d_main_trace, d_aux_trace, n_rows, challenges):
    """d_aux_trace: DeviceBuffer of [>= 90][n_rows][3] words (columns 0..48 filled); challenges: 63 XFE (Montgomery words)."""
    ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64).reshape(NUM_CHALLENGES, 3))
    if d_main_trace.n_words < NUM_MAIN_COLUMNS * n_rows or d_aux_trace.n_words < NUM_AUX_COLUMNS * n_rows * 3:
        raise ValueError("the traces need 379 main and at least 90 auxiliary columns")
    ctx._check(ctx.lib.tvm_fill_derived_aux_columns(ctx.handle, d_main_trace.ptr, d_aux_trace.ptr, n_rows,
                                                    ch.ctypes.data_as(C.c_void_p)), "tvm_fill_derived_aux_columns")
