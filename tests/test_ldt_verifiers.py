"""The device low-degree-test provers, tested the way the reference tests its own (fri.rs:990-1426, stir.rs:1528-2003):
prove, then verify with a restatement of the reference's verifier (oracle/ldt_verifier.py).  Honest low-degree
codewords are accepted and the verifier derives the prover's first-round indices; codewords of too high a degree and
corrupted transcripts are rejected."""
import numpy as np
import pytest

from oracle import ldt_verifier as lv
from triton_vm_amd import ArithmeticDomain, field, stark
from triton_vm_amd.prover import ProofStream

from .test_stir import small_stir


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


def fri_setup(orc, rng, log2_degree_bound, log2_expansion, high_degree=False):
    domain = ArithmeticDomain.of_length(1 << (log2_degree_bound + log2_expansion)).with_offset(field.generator())
    n_coeffs = domain.length if high_degree else 1 << log2_degree_bound
    poly = orc.random_elements(rng, (n_coeffs, 3))
    return domain, orc.coset_evaluate(poly, odom(orc, domain), 3).reshape(-1, 3)


@pytest.mark.parametrize("log2_bound,rounds,checks", [(4, 2, 3), (6, 3, 5), (3, 0, 2)])
def test_fri_prove_then_verify(ctx, orc, log2_bound, rounds, checks):
    rng = np.random.default_rng(log2_bound)
    domain, codeword = fri_setup(orc, rng, log2_bound, 2)
    ps = ProofStream(ctx.lib)
    a_indices, *_ = stark.fri_prove(ctx, domain, rounds, checks, ctx.to_device(codeword), ps)
    max_degree = ((1 << log2_bound) - 1) >> rounds
    assert lv.fri_verify(ps.verifier_view(), odom(orc, domain), rounds, checks, max_degree) == a_indices


def test_fri_rejects_high_degree_and_corruption(ctx, orc):
    rng = np.random.default_rng(1)
    domain, codeword = fri_setup(orc, rng, 5, 2, high_degree=True)
    ps = ProofStream(ctx.lib)
    stark.fri_prove(ctx, domain, 2, 4, ctx.to_device(codeword), ps)
    with pytest.raises(lv.VerificationError):
        lv.fri_verify(ps.verifier_view(), odom(orc, domain), 2, 4, 31 >> 2)
    # an honest proof with one revealed leaf, one authentication node or the last codeword tampered with
    domain, codeword = fri_setup(orc, rng, 5, 2)
    ps = ProofStream(ctx.lib)
    stark.fri_prove(ctx, domain, 2, 4, ctx.to_device(codeword), ps)
    lv.fri_verify(ps.verifier_view(), odom(orc, domain), 2, 4, 31 >> 2)
    for victim in ("fri response 0", "fri auth 1", "fri last codeword", "fri root 1"):
        bad = ProofStream(ctx.lib)
        bad.log = [(n, pl.copy(), fs) for n, pl, fs in ps.log]
        k = next(i for i, (n, _, _) in enumerate(bad.log) if n == victim)
        bad.log[k][1].reshape(-1)[0] ^= np.uint64(1)
        with pytest.raises(lv.VerificationError):
            lv.fri_verify(bad.verifier_view(), odom(orc, domain), 2, 4, 31 >> 2)


@pytest.mark.parametrize("log2_bound,queries", [(6, [(3, 1), (2, 0)]), (8, [(5, 2), (3, 1), (4, 0)]), (4, [(3, 0)])])
def test_stir_prove_then_verify(ctx, orc, log2_bound, queries):
    rng = np.random.default_rng(log2_bound)
    stir = small_stir(log2_bound, queries)
    poly = orc.random_elements(rng, (1 << log2_bound, 3))
    codeword = orc.coset_evaluate(poly, odom(orc, stir.initial_domain), 3).reshape(-1, 3)
    ps = ProofStream(ctx.lib)
    first = stir.prove(ctx, ctx.to_device(codeword), ps)
    assert lv.stir_verify(ps.verifier_view(), stir) == first


def test_stir_rejects_high_degree_and_corruption(ctx, orc):
    rng = np.random.default_rng(3)
    stir = small_stir(6, [(3, 1), (2, 0)])
    # too high a degree: a polynomial with as many coefficients as the domain has points
    poly = orc.random_elements(rng, (stir.initial_domain.length, 3))
    codeword = orc.coset_evaluate(poly, odom(orc, stir.initial_domain), 3).reshape(-1, 3)
    ps = ProofStream(ctx.lib)
    stir.prove(ctx, ctx.to_device(codeword), ps)
    with pytest.raises(lv.VerificationError):
        lv.stir_verify(ps.verifier_view(), stir)
    poly = orc.random_elements(rng, (1 << 6, 3))
    codeword = orc.coset_evaluate(poly, odom(orc, stir.initial_domain), 3).reshape(-1, 3)
    ps = ProofStream(ctx.lib)
    stir.prove(ctx, ctx.to_device(codeword), ps)
    lv.stir_verify(ps.verifier_view(), stir)
    for victim in ("stir response leafs", "stir response auth", "stir ood values", "stir final polynomial"):
        bad = ProofStream(ctx.lib)
        bad.log = [(n, pl.copy(), fs) for n, pl, fs in ps.log]
        k = next(i for i, (n, pl, _) in enumerate(bad.log) if n == victim and pl.size)
        bad.log[k][1].reshape(-1)[0] ^= np.uint64(1)
        with pytest.raises(lv.VerificationError):
            lv.stir_verify(bad.verifier_view(), stir)
