"""csrc/extend.hip (tvm_extend_aux_table: the auxiliary table's 49 cross-table-argument columns as prefix scans of
affine maps) against the oracle's row-by-row restatement of the reference's `extend` (oracle/vm/tables.py), on the
valid traces of tests/vm_fixture.py -- on the CPU fiber emulation (-m "not gpu") and on the MI355X (-m gpu)."""
import numpy as np
import pytest

from tests import vm_fixture as vf
from triton_vm_amd import master_table as mtab


@pytest.mark.parametrize("which", ["tiny", "every"])
def test_extend_matches_oracle(ctx, which):
    main, aux, ch, _ = vf.valid_tables(which)
    n = main.shape[1]
    d_main = ctx.to_device(main)
    start = aux.copy()
    start[:90] = 0                                             # only the randomizer column is given
    d_aux = ctx.to_device(start)
    mtab.extend(ctx, d_main, d_aux, n, ch)
    got = d_aux.download((91, n, 3))
    for c in range(91):
        assert (got[c] == aux[c]).all(), f"aux column {c}"


def test_extend_with_many_scan_tiles(ctx, orc):
    """A loop program of a few thousand cycles: the scan runs over several 1024-row tiles (carry between workgroups)."""
    from oracle import degree_lowering as dlo
    from oracle.vm import isa, tables as T, vm

    program = isa.parse("""
        push 1500 call loop pop 1
        sponge_init push 0 push 0 push 0 push 0 push 0 push 0 push 0 push 0 push 0 push 0 sponge_absorb sponge_squeeze
        pop 5 pop 5 halt
        loop: dup 0 push 0 eq skiz return
              dup 0 dup 0 push 100 add write_mem 1 pop 1
              push 120 read_mem 1 pop 2
              dup 0 split lt pop 1
              push -1 add recurse
    """)
    aet, output = vm.trace_execution(program)
    mt = T.MasterMainTable(aet).pad()
    n = mt.padded_height
    assert n >= 4096 * 4
    rng = np.random.default_rng(8)
    sampled = [[int(v) for v in rng.integers(0, T.P, 3, dtype=np.uint64)] for _ in range(59)]
    challenges = T.derive_challenges(sampled, vm.hash_varlen(program.to_bwords()), [], output)
    want = orc.to_mont(np.array(T.extend(mt.tables, challenges), dtype=object))
    main = np.zeros((379, n), np.uint64)
    main[:149] = orc.to_mont(np.array(mt.columns(), dtype=object))
    ch = orc.to_mont(np.array(challenges, dtype=object))
    d_main, d_aux = ctx.to_device(main), ctx.to_device(np.zeros((91, n, 3), np.uint64))
    ctx._check(ctx.lib.tvm_extend_aux_table(ctx.handle, d_main.ptr, d_aux.ptr, n, ch.ctypes.data), "tvm_extend_aux_table")
    got = d_aux.download((91, n, 3))
    for c in range(49):
        assert (got[c] == want[c]).all(), f"aux column {c}"
    assert not got[49:].any()
