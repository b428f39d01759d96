"""The whole hot path of Prover::prove (triton_vm_amd/prover.py, step order of stark.rs:331-719) against
an oracle-only re-computation with the same challenges, and against the protocol's own invariant: the
combination codeword handed to FRI is a low-degree codeword, so the last FRI polynomial is short."""
import numpy as np
import pytest

from triton_vm_amd import field
from triton_vm_amd.prover import Prover, StarkParameters, derive_challenges


def start_of_verification(prover):
    """Verifier::verify's first steps (stark.rs:1388-1420): the claim goes into the sponge, the padded height is read"""
    view = prover.transcript.verifier_view()
    view.alter_fiat_shamir_state_with(prover.claim.encode())
    assert field.from_mont(int(view.dequeue("log2 padded height")[0])) == prover.p.padded_height.bit_length() - 1
    return view


def odom(orc, d):
    return orc.Domain(d.offset, d.generator, d.length)


def xsum(orc, terms):
    acc = np.zeros(3, np.uint64)
    for t in terms:
        acc = orc.xfe_add(acc, t)
    return acc


def test_hot_path_matches_oracle_recomputation(ctx, orc):
    rng = np.random.default_rng(77)
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=2)
    n, h = p.trace.length, p.h
    assert (n, p.ldt.length, p.quotient.length) == (16, 128, 128)
    main_trace = orc.random_elements(rng, (379, n))
    aux_trace = orc.random_elements(rng, (91, n, 3))
    prover = Prover(ctx, p, main_trace, aux_trace, seed=5)
    prover.capture = {}
    prover.prove()
    c = prover.capture
    main_rnd = prover.main.d_randomizers.download((379, h))
    aux_rnd = prover.aux.d_randomizers.download((91, h, 3))

    ldt, quot, trace = odom(orc, p.ldt), odom(orc, p.quotient), odom(orc, p.trace)
    main_lde = orc.lde_table(main_trace, main_rnd, ldt, 1)
    aux_lde = orc.lde_table(aux_trace, aux_rnd, ldt, 3)
    assert (orc.merkle_tree(orc.hash_rows(main_lde))[1] == c["main_root"]).all()
    assert (orc.merkle_tree(orc.hash_rows(aux_lde.reshape(128, -1)))[1] == c["aux_root"]).all()

    q = orc.quotients_combined(main_lde, aux_lde, trace, quot, c["challenges"], c["quotient_weights"])
    seg = orc.interpolate_quotient_segments(q, quot)
    polys, seg_cws = orc.randomize_quotient_segments(seg, prover.quotient_randomizer, ldt)
    assert (orc.merkle_tree(orc.hash_rows(seg_cws.reshape(128, 15)))[1] == c["quot_root"]).all()

    alpha = c["alpha"]
    alpha_next = np.array([orc.lib().orc_bfe_mul(int(x), p.trace.generator) for x in alpha], np.uint64)
    assert (c["ood_main"][0] == orc.out_of_domain_row(main_trace, main_rnd, alpha, 1)).all()
    assert (c["ood_aux"][1] == orc.out_of_domain_row(aux_trace, aux_rnd, alpha_next, 3)).all()

    # combination + DEEP (stark.rs:508-625)
    wm, wq, wd = c["weights_ma"], c["weights_q"], c["weights_d"]
    comb = orc.weighted_sum_of_columns(main_trace, main_rnd, wm[:379], 1)
    comb_aux = orc.weighted_sum_of_columns(aux_trace, aux_rnd, wm[379:], 3)
    comb = np.array([orc.xfe_add(a, b) for a, b in zip(comb, comb_aux)], np.uint64)
    ma_cw = orc.coset_evaluate(comb, ldt, 3).reshape(-1, 3)
    a4 = orc.xfe_pow(alpha, 4)
    za4 = orc.xfe_pow(np.array([orc.lib().orc_bfe_mul(int(x), orc.bfe(3)) for x in alpha], np.uint64), 4)
    p_poly = np.array([xsum(orc, [orc.xfe_mul(wq[k], polys[k, j]) for k in range(4)]) for j in range(polys.shape[1])])
    r_poly = np.array([xsum(orc, [orc.xfe_mul(wq[k], polys[k, j]) for k in range(1, 5)]) for j in range(polys.shape[1])])
    p_cw = orc.coset_evaluate(p_poly, ldt, 3).reshape(-1, 3)
    r_cw = orc.coset_evaluate(r_poly, ldt, 3).reshape(-1, 3)
    parts = [orc.deep_codeword(ma_cw, ldt, alpha, orc.poly_eval_xfe(comb, alpha)),
             orc.deep_codeword(ma_cw, ldt, alpha_next, orc.poly_eval_xfe(comb, alpha_next)),
             orc.deep_codeword(p_cw, ldt, a4, orc.poly_eval_xfe(p_poly, a4)),
             orc.deep_codeword(r_cw, ldt, za4, orc.poly_eval_xfe(r_poly, za4))]
    want = np.array([xsum(orc, [orc.xfe_mul(parts[k][i], wd[k]) for k in range(4)]) for i in range(128)])
    assert (c["combination"] == want).all()

    # the transcript end to end, the way a verifier reads it: replay the Fiat-Shamir schedule of stark.rs:1386-1530 up to
    # the low-degree test, let the restated FRI verifier (oracle/ldt_verifier.py) accept it, and authenticate the
    # opened rows against the three table roots
    from oracle import ldt_verifier as lv

    view = start_of_verification(prover)
    assert (view.dequeue("main root") == c["main_root"]).all()
    assert (derive_challenges(prover.ctx.lib, view.sample_scalars(59), prover.claim) == c["challenges"]).all()
    view.dequeue("aux root")
    view.sample_scalars(1)
    view.dequeue("quot root")
    assert (view.sample_scalars(1)[0] == alpha).all()
    for name in ("ood main", "ood aux", "ood main next", "ood aux next", "ood quot p", "ood quot r"):
        view.dequeue(name)
    view.sample_scalars(3)
    max_degree = (p.randomized_trace_len - 1) >> p.fri_rounds
    opened_at = lv.fri_verify(view, ldt, p.fri_rounds, p.num_collinearity_checks, max_degree)
    for name, root, width in (("main", c["main_root"], 379), ("aux", c["aux_root"], 273), ("quot", c["quot_root"], 15)):
        rows = np.asarray(view.dequeue(f"{name} rows"), np.uint64).reshape(len(opened_at), width)
        auth = np.asarray(view.dequeue(f"{name} auth"), np.uint64).reshape(-1, 5)
        lv.verify_inclusion(root, p.ldt.length, opened_at, orc.hash_rows(rows), auth)
    assert not view.pending
    assert (np.asarray(prover.opened["main"]).reshape(len(opened_at), 379) == main_lde[opened_at]).all()
    # stark.rs:2367-2398: proving leaves the trace tables as they were
    assert (prover.main.d_trace.download(main_trace.shape) == main_trace).all()
    assert (prover.aux.d_trace.download(aux_trace.shape) == aux_trace).all()

    # FRI: the last polynomial respects the degree bound of a randomized_trace_len-degree input
    bound = p.randomized_trace_len >> p.fri_rounds
    assert (prover.last_polynomial[bound:] == 0).all()
    assert prover.last_polynomial[:bound].any()


def test_log_blowup_4_quotient_domain_is_the_short_domain(ctx, orc):
    """BASELINE config 5's shape (FRI expansion factor 16): the quotient domain is then shorter than the LDT domain, the
    tables are evaluated on the LDT domain and the AIR reads a stride view of them, the DEEP codeword is built on the
    quotient domain and low-degree-extended to the LDT domain (stark.rs:501-506, 629-639).  The transcript must pass
    the restated FRI verifier and the opened rows must authenticate against the roots."""
    from oracle import ldt_verifier as lv

    rng = np.random.default_rng(16)
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=3, log2_expansion=4)
    assert p.quotient.length == 128 and p.ldt.length == 512
    main_trace, aux_trace = orc.random_elements(rng, (379, p.trace.length)), orc.random_elements(rng, (91, p.trace.length, 3))
    prover = Prover(ctx, p, main_trace, aux_trace, seed=6)
    prover.capture = {}
    prover.prove()
    c = prover.capture
    # the quotient codeword is the oracle's, evaluated on the quotient domain from tables extended onto the LDT domain
    main_rnd, aux_rnd = prover.main.d_randomizers.download((379, p.h)), prover.aux.d_randomizers.download((91, p.h, 3))
    quot, trace = odom(orc, p.quotient), odom(orc, p.trace)
    main_q, aux_q = orc.lde_table(main_trace, main_rnd, quot, 1), orc.lde_table(aux_trace, aux_rnd, quot, 3)
    q = orc.quotients_combined(main_q, aux_q, trace, quot, c["challenges"], c["quotient_weights"])
    seg = orc.interpolate_quotient_segments(q, quot)
    _polys, seg_cws = orc.randomize_quotient_segments(seg, prover.quotient_randomizer, odom(orc, p.ldt))
    assert (orc.merkle_tree(orc.hash_rows(seg_cws.reshape(512, 15)))[1] == c["quot_root"]).all()
    # low degree after folding, verifier replay
    bound = p.randomized_trace_len >> p.fri_rounds
    assert (prover.last_polynomial[bound:] == 0).all() and prover.last_polynomial[:bound].any()
    view = start_of_verification(prover)
    roots = {"main": view.dequeue("main root")}
    view.sample_scalars(59)
    roots["aux"] = view.dequeue("aux root")
    view.sample_scalars(1)
    roots["quot"] = view.dequeue("quot root")
    view.sample_scalars(1)
    for name in ("ood main", "ood aux", "ood main next", "ood aux next", "ood quot p", "ood quot r"):
        view.dequeue(name)
    view.sample_scalars(3)
    opened_at = lv.fri_verify(view, odom(orc, p.ldt), p.fri_rounds, p.num_collinearity_checks, bound - 1)
    for name, width in (("main", 379), ("aux", 273), ("quot", 15)):
        rows = np.asarray(view.dequeue(f"{name} rows"), np.uint64).reshape(len(opened_at), width)
        auth = np.asarray(view.dequeue(f"{name} auth"), np.uint64).reshape(-1, 5)
        lv.verify_inclusion(roots[name], p.ldt.length, opened_at, orc.hash_rows(rows), auth)
    assert not view.pending


@pytest.mark.gpu
def test_full_size_pipeline_low_degree_invariant(orc):
    """BASELINE config 1 (2^20 padded rows, 652 words per row) on the MI355X: the size-independent
    invariant -- whatever the tables hold, the codeword handed to FRI is low degree, so after all
    folding rounds the last polynomial has at most randomized_trace_len >> rounds coefficients -- and the
    transcript read the way a verifier reads it: the restated FRI verifier accepts it (173 queries over 13
    rounds of the 2^23-point domain) and the 173 opened rows of each table authenticate against its root."""
    from oracle import ldt_verifier as lv
    from triton_vm_amd import Context

    ctx = Context(0)
    p = StarkParameters(20)
    prover = Prover(ctx, p, seed=3)
    prover.prove()
    bound = p.randomized_trace_len >> p.fri_rounds
    assert prover.last_codeword.shape[0] == p.ldt.length >> p.fri_rounds
    assert (prover.last_polynomial[bound:] == 0).all() and prover.last_polynomial[:bound].any()

    view = start_of_verification(prover)
    roots = {"main": view.dequeue("main root")}
    view.sample_scalars(59)
    roots["aux"] = view.dequeue("aux root")
    view.sample_scalars(1)
    roots["quot"] = view.dequeue("quot root")
    view.sample_scalars(1)
    for name in ("ood main", "ood aux", "ood main next", "ood aux next", "ood quot p", "ood quot r"):
        view.dequeue(name)
    view.sample_scalars(3)
    opened_at = lv.fri_verify(view, odom(orc, p.ldt), p.fri_rounds, p.num_collinearity_checks, bound - 1)
    assert len(opened_at) == p.num_collinearity_checks
    for name, width in (("main", 379), ("aux", 273), ("quot", 15)):
        rows = np.asarray(view.dequeue(f"{name} rows"), np.uint64).reshape(len(opened_at), width)
        auth = np.asarray(view.dequeue(f"{name} auth"), np.uint64).reshape(-1, 5)
        lv.verify_inclusion(roots[name], p.ldt.length, opened_at, orc.hash_rows(rows), auth)
    assert not view.pending
    ctx.close()
