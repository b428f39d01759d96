"""The restated STIR / Reed-Solomon parameter arithmetic (triton_vm_amd/low_degree_test.py) against every constant the
reference pins: the q-ary entropy table (low_degree_test/mod.rs:406-423), the field-size constant (mod.rs:395-401) and
the two worked examples of the over-sampling bound (stir.rs:739-747)."""
import math

import pytest

from triton_vm_amd import low_degree_test as ldt


def test_q_ary_entropy_table():
    want = {1: 0.505208333333361, 2: 0.254225406898247, 3: 0.127831064808346, 4: 0.064256719096972,
            5: 0.032294907939134, 6: 0.016229766017215, 7: 0.008155804230956, 8: 0.004098304720073}
    for log2_expansion, value in want.items():
        assert abs(ldt.ReedSolomonCode(log2_expansion).q_ary_entropy() - value) < 1e-4


def test_log2_extension_field_size():
    p = 2**64 - 2**32 + 1
    assert abs(ldt.LOG2_FIELD_SIZE_F - 3 * math.log2(p)) < 1e-4


def test_oversampling_worked_examples():
    params = ldt.StirParameters(160, 2, 20)
    assert params.num_total_in_domain_queries(23, 160) == 184   # lambda = 160, U = 2^23, k = 160
    assert params.num_total_in_domain_queries(8, 160) == 610    # lambda = 160, U = 2^8,  k = 160


def test_setup_invariants():
    """what try_into_stir guarantees (stir.rs:420-560) and Stark::stir searches for (stark.rs:1972-2032)"""
    for log2_height in (10, 16, 20, 22):
        stir = ldt.stark_stir(1 << log2_height)
        h = stir.num_trace_randomizers()
        assert stir.initial_domain.length >= ldt.randomized_trace_len(1 << log2_height, h) * 4
        assert stir.folding_factor == 4 and stir.final_degree > 0
        degree = (stir.initial_domain.length // 4 - 1) // 4
        for in_domain, out_of_domain in stir.round_queries:
            assert in_domain + out_of_domain <= degree // 4   # the quotient never collapses to zero
            degree //= 4
        assert degree == stir.final_degree


def test_invalid_parameters_are_rejected():
    with pytest.raises(ldt.LdtParameterError):
        ldt.StirParameters(42, 1, 11, log2_folding_factor=1).try_into_stir()
    with pytest.raises(ldt.LdtParameterError):
        ldt.StirParameters(42, 0, 11).try_into_stir()
    with pytest.raises(ldt.LdtParameterError):
        ldt.StirParameters(42, 1, 1).try_into_stir()
    with pytest.raises(ldt.LdtParameterError):
        ldt.StirParameters(42, 2, 40).try_into_stir()
