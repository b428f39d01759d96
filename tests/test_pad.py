"""csrc/pad.hip (tvm_pad_main_table: MasterMainTable::pad's nine table-specific rules) against the oracle's restatement
(oracle/vm/tables.py), on real traces from the oracle-side VM; then pad + degree-lowering fill + extend entirely on the
device must reproduce the oracle's padded, extended tables -- on which the pinned AIR vanishes (tests/test_vm_tables.py)."""
import numpy as np
import pytest

from tests import vm_fixture as vf
from triton_vm_amd import master_table as mtab


def _unpadded(orc, which):
    from oracle.vm import tables as T

    _, aet, _, _ = vf.run(which)
    mt = T.MasterMainTable(aet)
    n = mt.padded_height
    cols = np.zeros((379, n), np.uint64)
    c = 0
    for t in T.TABLES:
        rows = mt.tables[t]
        for k in range(T.MAIN_WIDTH[t]):
            if rows:
                cols[c, :len(rows)] = orc.to_mont(np.array([r[k] % T.P for r in rows], dtype=object))
            c += 1
    return cols, [mt.lengths[t] for t in mtab.TABLE_ORDER], n


@pytest.mark.parametrize("which", ["tiny", "every"])
def test_pad_and_fill_match_oracle(ctx, orc, which):
    main, aux, ch, _ = vf.valid_tables(which)           # the oracle's padded + derived + extended tables
    unpadded, lengths, n = _unpadded(orc, which)
    assert n == main.shape[1]
    d_main = ctx.to_device(unpadded)
    mtab.pad(ctx, d_main, n, lengths)
    got = d_main.download((379, n))
    for c in range(379):
        assert (got[c] == main[c]).all(), f"main column {c}"
    # ... and the whole host `gen` tail on the device: pad -> fill -> extend -> fill
    start = np.zeros((91, n, 3), np.uint64)
    start[90] = aux[90]
    d_aux = ctx.to_device(start)
    mtab.extend(ctx, d_main, d_aux, n, ch)
    assert (d_aux.download((91, n, 3)) == aux).all()


def test_pad_argument_checks(ctx):
    from triton_vm_amd.capi import TritonHipError

    d = ctx.alloc(379 * 8)
    with pytest.raises(TritonHipError):
        mtab.pad(ctx, d, 8, [9, 2, 0, 0, 2, 0, 0, 0, 0])       # a table longer than the padded height
    with pytest.raises(ValueError):
        mtab.pad(ctx, d, 8, [1, 2, 3])
