"""The C++ host side (triton_vm_amd/host/triton_host.cpp: ArithmeticDomain, MasterTable, ProofStream, Prover::prove over
the C ABI) against the Python mirror of the same reference code: identical proofs, word for word."""
import os

import numpy as np
import pytest

from triton_vm_amd import native_host
from triton_vm_amd.prover import Claim, Prover, StarkParameters


def _host_library(ctx):
    backend = ctx.lib._name
    if ctx.kind == "emu":
        return native_host.load_host_library(backend, os.path.join(os.path.dirname(backend), "libtriton_host_emu.so"))
    return native_host.load_host_library(backend)


@pytest.mark.parametrize("log2_rows,h,checks,log2_expansion", [(3, 3, 2, 2), (4, 5, 4, 2), (3, 3, 3, 4)])
def test_cpp_prover_proof_equals_python_prover_proof(ctx, orc, log2_rows, h, checks, log2_expansion):
    if (log2_rows, log2_expansion) != (3, 2) and ctx.kind == "emu":
        pytest.skip("one case on the emulation (CPU suite time); all on the GPU")
    rng = np.random.default_rng(log2_rows)
    p = StarkParameters(log2_rows, num_trace_randomizers=h, num_collinearity_checks=checks, log2_expansion=log2_expansion)
    n = p.trace.length
    main_trace, aux_trace = orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))
    claim = Claim(orc.random_elements(rng, 5), orc.random_elements(rng, 3), orc.random_elements(rng, 2))
    py = Prover(ctx, p, main_trace, aux_trace, seed=9, claim=claim)
    want = py.prove().proof().words
    native = native_host.NativeProver(ctx, _host_library(ctx), p, py.main.d_trace, py.main.d_randomizers, py.aux.d_trace,
                                      py.aux.d_randomizers, py.quotient_randomizer, claim)
    got = native.prove()
    assert got.size == want.size and (got == want).all()
    # and a second run on the same object reproduces it (nothing is left behind in the context)
    assert (native.prove() == got).all()


def test_cpp_host_proves_the_reference_snapshot_from_the_execution_trace(ctx, orc):
    """triton_vm::prove_execution -- the C++ host's Prover::prove(claim, aet): fill, pad, extend, the seeded randomness, the
    hot path and the transcript -- on the program, claim and seed of the reference's proof-digest snapshot (proof.rs:200-226)"""
    from tests import test_proof_snapshot as snap
    from tests import vm_fixture as vf
    from tests.test_fill import aet_arrays
    from triton_vm_amd.proof_stream import Proof

    program, aet, public_input, output = vf.run("tiny")
    words = native_host.prove_execution(ctx, _host_library(ctx), aet_arrays(orc, aet), aet.padded_height(),
                                        snap.claim_of(orc, program, public_input, output), snap.prover_seed(snap.SEED_U64))
    assert Proof(words).digest(ctx.lib) == snap.SNAPSHOT


def test_cpp_host_reports_errors(ctx):
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=2)
    lib = _host_library(ctx)
    bad = native_host.NativeProver(ctx, lib, p, ctx.alloc(8), ctx.alloc(8), ctx.alloc(8), ctx.alloc(8),
                                   np.zeros((p.num_quotient_randomizers, 3), np.uint64))
    bad.bufs = (type("Null", (), {"ptr": None})(),) * 4   # null traces: the C ABI refuses, the C++ host reports it
    with pytest.raises(RuntimeError, match="tvm_lde_table"):
        bad.prove()
