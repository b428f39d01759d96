"""The restated AIR on VALID traces -- the reference's strongest test of its own constraints
(`triton_constraints_evaluate_to_zero`, /root/reference/triton-vm/src/stark.rs:2899-3016, 4186-4255, 4820-4823): the
oracle-side VM (oracle/vm/vm.py) executes a program, oracle/vm/tables.py fills, pads and extends the master tables, and
all 604 constraints of the fingerprint-pinned circuit (tests/test_air_fingerprint.py) must vanish: initial constraints
on the first row, consistency constraints on every row, transition constraints on every pair of consecutive rows,
terminal constraints (incl. the cross-table argument) on the last row.  This checks the VM / fill / pad / extend
restatement against the pinned AIR, and is the oracle for the device-side `extend` (tests/test_extend.py)."""
import numpy as np
import pytest

from tests import vm_fixture as vf


def test_vm_executes_every_instruction():
    """stark.rs:4805-4826"""
    from oracle.vm import isa

    _, aet, _, output = vf.run("every")
    assert {r[3] for r in aet.processor_trace} == {op for op, _ in isa.INSTRUCTIONS.values()}
    assert len(output) == 5


@pytest.mark.parametrize("which", ["tiny", "every"])
def test_constraints_vanish_on_valid_traces(which):
    main, aux, ch, mt = vf.valid_tables(which)
    assert main.shape[1] == mt.padded_height and (main.shape[1] & (main.shape[1] - 1)) == 0
    assert vf.constraint_violations(main, aux, ch) == []


def test_a_corrupted_cell_is_caught():
    """the test above is not vacuous: one changed cell in a non-derived column violates constraints near that row"""
    main, aux, ch, _ = vf.valid_tables("tiny")
    rng = np.random.default_rng(3)
    for col, table in ((17, main), (40, main), (5, aux), (30, aux)):
        t = table.copy()
        row = int(rng.integers(1, 6))
        t[col, row] = t[col, row] + np.uint64(1) if table is main else t[col, row] + np.array([1, 0, 0], np.uint64)
        m, a = (t, aux) if table is main else (main, t)
        assert vf.constraint_violations(m, a, ch, rows=range(0, 8)) != [], (col, row)


def test_table_heights_of_the_every_instruction_program():
    _, aet, _, _ = vf.run("every")
    heights = {t: aet.height_of_table(t) for t in ("Program", "Processor", "OpStack", "Ram", "JumpStack", "Hash",
                                                   "Cascade", "Lookup", "U32")}
    assert heights["Processor"] == heights["JumpStack"] and heights["Lookup"] == 256
    assert aet.padded_height() == 2048                      # the Cascade table dominates (1868 distinct 16-bit limbs)
