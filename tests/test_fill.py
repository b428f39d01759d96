"""csrc/fill_aet.hip (tvm_fill_main_table: the master main table's fills from the AET -- stable radix sorts of the
memory-like tables, clock-jump-difference multiplicities, Bezout coefficients, u32 sections) against the oracle's
restatement (oracle/vm/tables.py) on AETs produced by the oracle-side VM; then the WHOLE host `gen` tail on the device:
fill -> pad -> degree-lowering fill -> extend -> degree-lowering fill equals the oracle's tables, on which the
reference-pinned AIR vanishes (tests/test_vm_tables.py)."""
import numpy as np
import pytest

from tests import vm_fixture as vf
from triton_vm_amd import master_table as mtab


def aet_arrays(orc, aet, host_bezout=True):
    """the oracle VM's AET in the layout of AlgebraicExecutionTrace (aet.rs:41-96): Montgomery words, row-major.
    host_bezout=False leaves the RAM table's Bezout coefficient polynomials to the device."""
    from oracle.vm import tables as T

    def M(rows, width):
        if not rows:
            return np.zeros((0, width), np.uint64)
        try:
            canonical = np.array(rows, dtype=np.uint64)       # the VM keeps canonical values: one pass, also at 2^20 rows
        except OverflowError:
            canonical = np.array([[v % T.P for v in r] for r in rows], dtype=object)
        return orc.to_mont(canonical.reshape(-1, width))

    hash_rows = lambda trace: [T.hash_table_row(0, ci, rnd, state) for ci, rnd, state in trace]   # Mode is set by fill
    b0 = b1 = []
    if host_bezout:
        ram_rows, _ = (T.fill_ram(aet) if aet.ram_trace else ([], []))
        unique = list(dict.fromkeys(r[T.M["Ram"]["RamPointer"]] for r in ram_rows))
        b0, b1 = T.bezout_coefficient_polynomials_coefficients(unique)
    arrays = {
        "program_words": orc.to_mont(np.array(aet.program.to_bwords(), dtype=object)),
        "instruction_multiplicities": np.array(aet.instruction_multiplicities, np.uint32),
        "processor_trace": M(aet.processor_trace, 39),
        "op_stack_trace": M(aet.op_stack_underflow_trace, 4),
        "ram_trace": M([list(r) + [0, 0, 0] for r in aet.ram_trace], 7),
        "bezout_coefficients_0": orc.to_mont(np.array(b0, dtype=object)) if b0 else np.zeros(0, np.uint64),
        "bezout_coefficients_1": orc.to_mont(np.array(b1, dtype=object)) if b1 else np.zeros(0, np.uint64),
        "program_hash_trace": M(hash_rows(aet.program_hash_trace), 67),
        "sponge_trace": M(hash_rows(aet.sponge_trace), 67),
        "hash_trace": M(hash_rows(aet.hash_trace), 67),
        "u32_entries": np.array([[T.OP[name], int(orc.to_mont([lhs])[0]), int(orc.to_mont([rhs])[0]), mult]
                                 for (name, lhs, rhs), mult in aet.u32_entries.items()], np.uint64).reshape(-1, 4),
        "cascade_entries": np.array([[limb, mult] for limb, mult in aet.cascade_multiplicities.items()], np.uint64).reshape(-1, 2),
        "lookup_multiplicities": np.array(aet.lookup_multiplicities, np.uint64),
    }
    if not host_bezout:
        del arrays["bezout_coefficients_0"], arrays["bezout_coefficients_1"]
    return arrays


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
