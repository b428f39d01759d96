"""The coset-wise ("just in time") formulation (triton_vm_amd/jit.py; reference: stark.rs:805-1006) must produce
exactly what the cached path produces: same roots, challenges, out-of-domain rows, combination codeword, last FRI
polynomial and opened rows, for every admissible number of passes."""
import numpy as np
import pytest

from triton_vm_amd.jit import JitProver
from triton_vm_amd.prover import Prover, StarkParameters

KEYS = ("main_root", "aux_root", "quot_root", "challenges", "alpha", "ood_main", "ood_aux", "combination")


def _capture(prover):
    prover.capture = {}
    prover.prove()
    return {k: np.array(prover.capture[k]) for k in KEYS} | {
        "last_polynomial": prover.last_polynomial, "main rows": prover.opened["main"], "aux rows": prover.opened["aux"]}


@pytest.mark.parametrize("passes", [1, 2, 8])
def test_jit_prover_equals_cached_prover(ctx, orc, passes):
    rng = np.random.default_rng(5)
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=4)
    n = p.trace.length
    main_trace, aux_trace = orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))
    want = _capture(Prover(ctx, p, main_trace, aux_trace, seed=11))
    got = _capture(JitProver(ctx, p, passes, main_trace, aux_trace, seed=11))
    for key, value in want.items():
        assert (got[key] == value).all(), key


def test_pass_count_must_divide_the_expansion(ctx):
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=4)
    with pytest.raises(ValueError):
        JitProver(ctx, p, 3)
