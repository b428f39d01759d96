"""csrc/fill_aet.hip (tvm_fill_main_table: the master main table's fills from the AET -- stable radix sorts of the
memory-like tables, clock-jump-difference multiplicities, Bezout coefficients, u32 sections) against the oracle's
restatement (oracle/vm/tables.py) on AETs produced by the oracle-side VM; then the WHOLE host `gen` tail on the device:
fill -> pad -> degree-lowering fill -> extend -> degree-lowering fill equals the oracle's tables, on which the
reference-pinned AIR vanishes (tests/test_vm_tables.py)."""
import numpy as np
import pytest

from tests import vm_fixture as vf
from triton_vm_amd import master_table as mtab


from oracle.vm.aet_export import aet_arrays  # noqa: E402,F401  (the tests import it from here)


@pytest.mark.parametrize("which,host_bezout", [("tiny", True), ("every", True), ("every", False)])
def test_fill_pad_extend_on_the_device_reproduce_the_oracle_tables(ctx, orc, which, host_bezout):
    from oracle.vm import tables as T

    main, aux, ch, mt = vf.valid_tables(which)
    _, aet, _, _ = vf.run(which)
    n = main.shape[1]
    d_main = ctx.alloc(379 * n)
    lengths = mtab.fill(ctx, d_main, n, aet_arrays(orc, aet, host_bezout))   # without them: the device's Bezout coefficients
    assert lengths == [mt.lengths[t] for t in mtab.TABLE_ORDER]
    # the unpadded fill, table by table
    unpadded = T.MasterMainTable(aet)
    got = d_main.download((379, n))
    c = 0
    for t in T.TABLES:
        rows = unpadded.tables[t]
        for k in range(T.MAIN_WIDTH[t]):
            want = np.zeros(n, np.uint64)
            if rows:
                want[:len(rows)] = orc.to_mont(np.array([r[k] % T.P for r in rows], dtype=object))
            assert (got[c] == want).all(), f"{t} column {k}"
            c += 1
    # ... and on through pad, the degree-lowering fills and extend
    mtab.pad(ctx, d_main, n, lengths)
    assert (d_main.download((379, n)) == main).all()
    start = np.zeros((91, n, 3), np.uint64)
    start[90] = aux[90]
    d_aux = ctx.to_device(start)
    mtab.extend(ctx, d_main, d_aux, n, ch)
    assert (d_aux.download((91, n, 3)) == aux).all()


def test_fill_argument_checks(ctx, orc):
    from triton_vm_amd.capi import TritonHipError

    _, aet, _, _ = vf.run("tiny")
    arrays = aet_arrays(orc, aet)
    d = ctx.alloc(379 * 64)
    with pytest.raises(TritonHipError):
        mtab.fill(ctx, d, 64, arrays)                      # the lookup table alone needs 256 rows
    with pytest.raises(ValueError):
        mtab.fill(ctx, d, 64, dict(arrays, lookup_multiplicities=np.zeros(3, np.uint64)))
