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


@pytest.mark.parametrize("passes", [2, 8])
def test_jit_prover_equals_cached_prover(ctx, orc, passes):
    if passes == 8 and ctx.kind == "emu":
        pytest.skip("the emulation runs the 2-pass case (CPU suite time); 8 passes, the reference's coset count, run on the GPU")
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


def test_out_of_memory_falls_back_to_the_coset_wise_path(ctx, orc):
    """master_table.rs:268-271: an extension that does not fit is not an error, the prover switches to the JIT path.
    The context's memory limit stands in for a full device; the proof must not change."""
    from triton_vm_amd import jit

    rng = np.random.default_rng(5)
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=4)
    n = p.trace.length
    main_trace, aux_trace = orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))
    want = {}
    prover, _ = jit.prove(ctx, p, main_trace, aux_trace, seed=11, capture=want)
    assert type(prover) is Prover
    prover.release()
    del prover
    ctx.trim()
    traces_bytes = 8 * (main_trace.size + aux_trace.size)
    full_tables = 8 * p.ldt.length * (379 + 273)
    # room for what is already held, the traces, and half of the extended tables (plus allocator granularity)
    ctx.set_memory_limit(ctx.memory_held() + traces_bytes + full_tables // 2 + (1 << 16))
    try:
        got = {}
        prover, _ = jit.prove(ctx, p, main_trace, aux_trace, seed=11, capture=got)
        assert isinstance(prover, JitProver) and prover.passes >= 2
    finally:
        ctx.set_memory_limit(0)
    for key in KEYS:
        assert (np.array(got[key]) == np.array(want[key])).all(), key


def test_proving_does_not_mutate_the_traces(ctx, orc):
    """stark.rs:2367-2398: computing the quotients (cached or coset-wise) leaves the trace tables as they were."""
    rng = np.random.default_rng(6)
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=4)
    n = p.trace.length
    main_trace, aux_trace = orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))
    # on the emulation only the coset-wise prover (the cached one is checked in tests/test_prover_pipeline.py)
    provers = [JitProver(ctx, p, 2, main_trace, aux_trace, seed=2)]
    if ctx.kind != "emu":
        provers.append(Prover(ctx, p, main_trace, aux_trace, seed=2))
    for prover in provers:
        prover.prove()
        assert (prover.main.d_trace.download(main_trace.shape) == main_trace).all()
        assert (prover.aux.d_trace.download(aux_trace.shape) == aux_trace).all()
