"""The device STIR prover (triton_vm_amd/low_degree_test.py over csrc/stir.hip; reference: stir.rs:885-993) against a
CPU restatement that does what the reference does -- coefficient-form polynomial arithmetic (oracle/stir_oracle.py) --
round by round with the randomness the prover sampled, and against the protocol's own invariant: an honest low-degree
codeword ends in a final polynomial of degree <= final_degree."""
import numpy as np
import pytest

from oracle import stir_oracle as so
from triton_vm_amd import field
from triton_vm_amd import low_degree_test as ldt
from triton_vm_amd.prover import ProofStream


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


def small_stir(log2_high_degree_bound, queries):
    """a STIR instance small enough for the coefficient-form oracle: the reference's structure (folding factor 4,
    domain halving, offsets) with hand-picked query counts"""
    params = ldt.StirParameters(8, 2, log2_high_degree_bound)
    stir = params.try_into_stir()
    stir.round_queries = queries[:-1]
    stir.final_num_in_domain_queries = queries[-1][0]
    stir.final_degree = ((1 << log2_high_degree_bound) - 1) // 4 ** (len(queries))
    return stir


def test_fold_polynomial_and_stacked_tree(ctx, orc):
    rng = np.random.default_rng(1)
    for n in (5, 16, 37):
        poly = orc.random_elements(rng, (n, 3))
        r = orc.random_elements(rng, 3)
        got, n_out = ldt.fold_polynomial(ctx, ctx.to_device(poly), n, 4, r)
        assert n_out == -(-n // 4)
        assert (got.download((n_out, 3)) == so.fold_polynomial(poly, 4, r)).all()
    for log_n, stack_height in ((3, 4), (6, 4), (5, 2), (11, 4)):
        cw = orc.random_elements(rng, (1 << log_n, 3))
        d_cw = ctx.to_device(cw)
        tree = ldt.StirMerkleTree(ctx, d_cw, 1 << log_n, stack_height)
        assert (tree.root() == so.stir_merkle_root(cw, stack_height)).all()
        leafs, _ = tree.inclusion_proof([0, tree.n_leaves - 1])
        want = so.stack(cw, stack_height)
        assert (leafs[0] == want[0]).all() and (leafs[1] == want[-1]).all()


def test_host_interpolation_matches_lagrange(ctx, orc):
    rng = np.random.default_rng(2)
    for k in (1, 2, 7):
        pts, vals = orc.random_elements(rng, (k, 3)), orc.random_elements(rng, (k, 3))
        out = np.empty((k, 3), np.uint64)
        assert ctx.lib.tvm_host_xfe_interpolate(pts.ctypes.data, vals.ctypes.data, k, out.ctypes.data) == 0
        want = so.lagrange_interpolate(list(pts), list(vals))
        want += [so.ZERO] * (k - len(want))
        assert (out == np.array(want)).all()
    pts = np.array([[1, 2, 3], [1, 2, 3]], np.uint64)
    assert ctx.lib.tvm_host_xfe_interpolate(pts.ctypes.data, pts.ctypes.data, 2, np.empty((2, 3), np.uint64).ctypes.data) != 0


def test_the_two_lagrange_forms_of_the_oracle_agree(orc):
    """oracle/stir_oracle.py: the O(k^2) form the oracle prover uses (basis polynomial = zerofier / (X - p_i)) against the cubic textbook
    form, base-field points lifted and extension-field points mixed as in a STIR quotient set"""
    rng = np.random.default_rng(11)
    for k in (1, 2, 5, 12):
        pts, vals = orc.random_elements(rng, (k, 3)), orc.random_elements(rng, (k, 3))
        pts[: k // 2, 1:] = 0
        a, b = so.lagrange_interpolate(list(pts), list(vals)), so.lagrange_interpolate_from_zerofier(list(pts), list(vals))
        assert len(so.poly_trim(a)) == len(so.poly_trim(b)) and all((x == y).all() for x, y in zip(so.poly_trim(a), so.poly_trim(b)))
        for p_, v_ in zip(pts, vals):
            assert (orc.poly_eval_xfe(np.array(b, np.uint64), p_) == v_).all()


@pytest.mark.parametrize("log2_bound,queries", [(6, [(3, 1), (2, 0)]), (8, [(5, 2), (3, 1), (4, 0)])])
def test_prover_rounds_match_the_coefficient_form_restatement(ctx, orc, log2_bound, queries):
    rng = np.random.default_rng(log2_bound)
    stir = small_stir(log2_bound, queries)
    domain = stir.initial_domain
    # an honest codeword: a polynomial of degree < 2^log2_bound on the initial domain
    poly = orc.random_elements(rng, (1 << log2_bound, 3))
    codeword = orc.coset_evaluate(poly, odom(orc, domain), 3).reshape(-1, 3)
    ps = ProofStream(ctx.lib)
    first = stir.prove(ctx, ctx.to_device(codeword), ps)
    assert len(first) == queries[0][0] and all(0 <= i < domain.length for i in first)

    cur = [c for c in poly]
    cur_domain = domain
    for rnd in stir.rounds:
        folded = so.fold_polynomial(np.array(cur + [so.ZERO] * (-len(cur) % 4)), 4, rnd["folding_randomness"])
        nxt_domain = rnd["domain"]
        assert nxt_domain.length == cur_domain.length // 2
        assert nxt_domain.offset == field.mont_mul(field.mont_mul(cur_domain.offset, cur_domain.offset), cur_domain.offset)
        evals = orc.coset_evaluate(folded, odom(orc, nxt_domain), 3).reshape(-1, 3)
        assert (so.stir_merkle_root(evals, 4) == rnd["root"]).all()
        for q, v in zip(rnd["ood_queries"], rnd["ood_values"]):
            assert (orc.poly_eval_xfe(folded, q) == v).all()
        folded_domain = cur_domain.pow(4)
        assert rnd["folded_queried"] == list(dict.fromkeys(i % folded_domain.length for i in rnd["queried_indices"]))
        for i, (p, v) in zip(rnd["folded_queried"], zip(rnd["quotient_set"], rnd["quotient_answers"])):
            assert p[0] == folded_domain.value(i) and p[1] == 0 and p[2] == 0
            assert (orc.poly_eval_xfe(folded, p) == v).all()
        cur = so.next_polynomial(folded, rnd["quotient_set"], rnd["quotient_answers"], rnd["degree_correction_randomness"])
        assert len(so.poly_trim(cur)) <= len(folded)  # the degree correction restores, never raises, the degree
        cur_domain = nxt_domain
    # the final round: one more fold, no quotienting; an honest codeword ends at degree <= final_degree
    want_final = so.fold_polynomial(np.array(cur + [so.ZERO] * (-len(cur) % 4)), 4, stir.final_folding_randomness)
    got_final = so.poly_trim(list(stir.final_polynomial))
    assert len(got_final) == len(so.poly_trim(list(want_final)))
    assert all((a == b).all() for a, b in zip(got_final, want_final))
    assert len(got_final) <= stir.final_degree + 1


@pytest.mark.parametrize("n,k,n_base", [(32, 5, 2), (128, 40, 38), (256, 33, 33)])
def test_next_polynomial_kernel_matches_long_division(ctx, orc, n, k, n_base):
    """tvm_stir_next_polynomial (pointwise on a coset + one interpolation) against interpolate / zerofier / long
    division / schoolbook product; from 32 points on the answer polynomial is evaluated on the coset by a transform instead of
    Horner's rule at every point"""
    rng = np.random.default_rng(9 + k)
    folded = orc.random_elements(rng, (n, 3))
    pts = orc.random_elements(rng, (k, 3))
    pts[:n_base, 1:] = 0  # leading base-field points, like queried domain values
    answers = np.array([orc.poly_eval_xfe(folded, p) for p in pts], np.uint64)
    r = orc.random_elements(rng, 3)
    ans_poly = np.empty((k, 3), np.uint64)
    assert ctx.lib.tvm_host_xfe_interpolate(pts.ctypes.data, answers.ctypes.data, k, ans_poly.ctypes.data) == 0
    from triton_vm_amd import ArithmeticDomain

    work = ArithmeticDomain.of_length(n).with_offset(field.mont_mul(field.generator(), field.generator()))
    out, d_folded = ctx.alloc(3 * n), ctx.to_device(folded)
    ctx._check(ctx.lib.tvm_stir_next_polynomial(ctx.handle, d_folded.ptr, n, pts.ctypes.data, ans_poly.ctypes.data, k,
                                                r.ctypes.data, work.c(), out.ptr), "next")
    want = so.next_polynomial(folded, pts, answers, r)
    want += [so.ZERO] * (n - len(want))
    assert (out.download((n, 3)) == np.array(want)).all()


def test_prove_with_stir_as_the_low_degree_test(ctx, orc):
    """Prover.prove with LdtChoice::Stir (the reference's automatic choice from 2^16 rows on): Stark::stir fixes the
    trace randomizers and the LDT domain; the combination codeword it hands to STIR is a low-degree codeword, so the
    final polynomial respects final_degree.  2^3 padded rows keeps the emulated AIR affordable; the security level is
    lowered so that STIR still has a full round at that size."""
    from triton_vm_amd.prover import Prover, StarkParameters

    p = StarkParameters(3, ldt="stir")
    stir = ldt.stark_stir(8, security_level=8)
    assert stir.round_queries, "the instance must exercise the quotienting round"
    p2 = StarkParameters(3, num_trace_randomizers=stir.num_trace_randomizers(), num_collinearity_checks=2)
    p2.stir = stir
    assert p2.ldt.length == stir.initial_domain.length
    prover = Prover(ctx, p2, seed=4)
    stream = prover.prove()
    # the proof with its StirResponse / StirOutOfDomainValues items decodes and re-encodes to itself
    from triton_vm_amd.proof_stream import ProofStream

    words = stream.proof().words
    decoded = ProofStream.from_proof(ctx.lib, words)
    assert (decoded.proof().words == words).all()
    assert [v for v, _ in decoded.items].count("stir response leafs") == len(stir.round_queries) + 1
    final = so.poly_trim(list(prover.last_polynomial))
    assert 0 < len(final) <= stir.final_degree + 1
    assert set(prover.opened) == {"main", "aux"} and prover.opened["main"].shape[0] == stir.num_first_round_queries()
    # the default-security instance exists and is consistent at this size too
    assert p.stir.initial_domain.length == p.ldt.length and p.h == p.stir.num_trace_randomizers()


@pytest.mark.parametrize("k", [1, 2, 7, 64, 204, 256, 300])
def test_device_interpolation_equals_host_interpolation(ctx, orc, k):
    """tvm_xfe_interpolate (one workgroup on the device; k > 256 falls through to the host form) against tvm_host_xfe_interpolate
    (Newton, itself checked against the oracle above): the same coefficients, and a repeated point is refused"""
    rng = np.random.default_rng(k)
    pts, vals = orc.random_elements(rng, (k, 3)), orc.random_elements(rng, (k, 3))
    pts[: k // 2, 1:] = 0      # as in STIR: the in-domain points are base-field elements, the out-of-domain ones are not
    want, got = np.empty((k, 3), np.uint64), np.empty((k, 3), np.uint64)
    assert ctx.lib.tvm_host_xfe_interpolate(pts.ctypes.data, vals.ctypes.data, k, want.ctypes.data) == 0
    assert ctx.lib.tvm_xfe_interpolate(ctx.handle, pts.ctypes.data, vals.ctypes.data, k, got.ctypes.data) == 0
    assert (got == want).all()
    if k >= 2:
        pts[k - 1] = pts[0]
        assert ctx.lib.tvm_xfe_interpolate(ctx.handle, pts.ctypes.data, vals.ctypes.data, k, got.ctypes.data) != 0


@pytest.mark.parametrize("pattern", ["all in the base field", "none", "interleaved", "extension points first", "one extension point last"])
@pytest.mark.parametrize("k", [33, 204])
def test_device_interpolation_takes_base_field_points_first_wherever_they_are(ctx, orc, k, pattern):
    """the device kernel multiplies the base-field points in first (cheaper steps) whatever their place in the input: the
    interpolant does not depend on the order, so every pattern gives the host form's coefficients"""
    rng = np.random.default_rng(1000 + k)
    pts, vals = orc.random_elements(rng, (k, 3)), orc.random_elements(rng, (k, 3))
    base = {"all in the base field": np.ones(k, bool), "none": np.zeros(k, bool), "interleaved": rng.integers(0, 2, k).astype(bool),
            "extension points first": np.arange(k) >= 2, "one extension point last": np.arange(k) < k - 1}[pattern]
    pts[base, 1:] = 0
    want, got = np.empty((k, 3), np.uint64), np.empty((k, 3), np.uint64)
    assert ctx.lib.tvm_host_xfe_interpolate(pts.ctypes.data, vals.ctypes.data, k, want.ctypes.data) == 0
    assert ctx.lib.tvm_xfe_interpolate(ctx.handle, pts.ctypes.data, vals.ctypes.data, k, got.ctypes.data) == 0
    assert (got == want).all()
