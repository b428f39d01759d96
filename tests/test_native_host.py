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


@pytest.mark.parametrize("log2_bound,queries", [(6, [(3, 1), (2, 0)]), (8, [(5, 2), (3, 1), (4, 0)]), (4, [(3, 0)])])
def test_cpp_stir_prover_equals_python_stir_prover(ctx, orc, log2_bound, queries):
    """Stir::prove of the C++ host (quotienting rounds included) against the Python host's: the same proof words"""
    from tests.test_ldt_verifiers import odom, small_stir
    from triton_vm_amd.prover import ProofStream

    rng = np.random.default_rng(log2_bound)
    stir = small_stir(log2_bound, queries)
    poly = orc.random_elements(rng, (1 << log2_bound, 3))
    d_codeword = ctx.to_device(orc.coset_evaluate(poly, odom(orc, stir.initial_domain), 3).reshape(-1, 3))
    ps = ProofStream(ctx.lib)
    want_first = stir.prove(ctx, d_codeword, ps)
    first, words = native_host.stir_prove(ctx, _host_library(ctx), stir, d_codeword)
    assert first == want_first
    assert (words == ps.proof().words).all()


def test_cpp_stir_parameters_and_whole_proof_equal_the_python_hosts(ctx, orc):
    """LdtChoice::Stir through tvmh_prove: Stark::stir's instance (derived in C++) and the proof equal the Python host's.
    On the emulation the default-security instance at this size has no quotienting round (covered above); on the GPU the
    test runs at 2^16 padded rows, where it has four."""
    from triton_vm_amd.low_degree_test import stark_stir

    host = _host_library(ctx)
    for log2_height, security_level, log2_expansion in ((3, 160, 2), (10, 160, 2), (16, 160, 2), (20, 160, 2), (22, 128, 3), (20, 80, 1), (24, 160, 4)):
        want = stark_stir(1 << log2_height, security_level=security_level, log2_ldt_expansion_factor=log2_expansion)
        got = native_host.stir_parameters(host, 1 << log2_height, security_level, log2_expansion)
        assert got == dict(domain_length=want.initial_domain.length, folding_factor=want.folding_factor, round_queries=want.round_queries,
                           final_num_in_domain_queries=want.final_num_in_domain_queries, final_degree=want.final_degree)
    if ctx.kind == "emu":
        return   # the whole proof with the derived instance: GPU (2^16 padded rows, four quotienting rounds)
    p = StarkParameters(16, ldt="stir")
    assert len(p.stir.round_queries) >= 3
    py = Prover(ctx, p, seed=12)
    want = py.prove().proof().words
    native = native_host.NativeProver(ctx, _host_library(ctx), p, py.main.d_trace, py.main.d_randomizers, py.aux.d_trace,
                                      py.aux.d_randomizers, py.quotient_randomizer)
    got = native.prove()
    assert got.size == want.size and (got == want).all()


def test_cpp_host_reports_errors(ctx):
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=2)
    lib = _host_library(ctx)
    bad = native_host.NativeProver(ctx, lib, p, ctx.alloc(8), ctx.alloc(8), ctx.alloc(8), ctx.alloc(8),
                                   np.zeros((p.num_quotient_randomizers, 3), np.uint64))
    bad.bufs = (type("Null", (), {"ptr": None})(),) * 4   # null traces: the C ABI refuses, the C++ host reports it
    with pytest.raises(RuntimeError, match="tvm_lde_table"):
        bad.prove()
