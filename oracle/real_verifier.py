"""TEST INFRASTRUCTURE (oracle): `Verifier::verify` (/root/reference/triton-vm/src/stark.rs:1388-1763) restated end to
end over a decoded proof, so that proofs of the device path can be put through the reference's acceptance procedure in a
container without cargo.  It reads a `triton_vm_amd.proof_stream.ProofStream.from_proof(..)` view (the decoding is the
product's mirror of `ProofStream::try_from(&Proof)`), evaluates the AIR with the reference-pinned circuit of the C oracle,
and runs the restated FRI verifier (oracle/ldt_verifier.py).  Anchors: it accepts the oracle prover's proofs whose digests
equal the reference's two snapshots (tests/test_proof_snapshot.py) -- proofs the reference's own verifier accepts -- and
rejects every single-word corruption tried.  Plain python over the C oracle.  The product never imports this file."""
import math

import numpy as np

from . import ldt_verifier as lv
from . import oracle as orc
from .vm import tables as T

NUM_MAIN, NUM_AUX, NUM_CONSTRAINTS = 379, 91, 604
SECTIONS = (("init", 0, 81), ("cons", 81, 178), ("tran", 178, 581), ("term", 581, 604))   # master_table.rs:1302-1359
VerificationError = lv.VerificationError


def _xsum(terms):
    acc = np.zeros(3, np.uint64)
    for t in terms:
        acc = orc.xfe_add(acc, t)
    return acc


def _lift(b):
    return np.array([int(b), 0, 0], np.uint64)


def _powers(x, n, first=0):
    return [orc.xfe_pow(x, first + i) for i in range(n)]


def verify(view, claim, security_level=160, log2_expansion=2, ldt_choice="fri"):
    """view: oracle/proof_decode.py's VerifierView of the proof (or any object with its four methods); claim: an object with
    program_digest / input / output (Montgomery words) and version; ldt_choice: "fri" or
    "stir" (Stark::ldt picks by padded height, stark.rs:1944-1951; the caller says which the prover used).
    Raises VerificationError; returns the first-round indices on acceptance."""
    from .real_prover import fri_num_collinearity_checks

    values = lambda a: [int(v) for v in orc.from_mont(np.asarray(a, np.uint64).reshape(-1))]
    # Claim (proof.rs:62-120) in the oracle's own BFieldCodec: the struct's fields last field first, a dynamically sized
    # one prefixed with its length (only the plain data of `claim` is read: digest, version, input, output)
    from .real_prover import Variant, enc_bfe_words, enc_struct, enc_vec_static

    encoded = enc_struct([enc_bfe_words(values(claim.program_digest)), enc_bfe_words([int(getattr(claim, "version", 6))]),
                          enc_vec_static(values(claim.input), len(claim.input)), enc_vec_static(values(claim.output), len(claim.output))],
                         Variant())
    view.alter_fiat_shamir_state_with(orc.to_mont(np.array(encoded.words, dtype=object)) if encoded.words else np.zeros(0, np.uint64))
    log2_padded_height = values(view.dequeue("Log2PaddedHeight"))[0]
    if log2_padded_height >= 32:
        raise VerificationError("Log2PaddedHeightTooLarge")
    padded_height = 1 << log2_padded_height
    # Stark::ldt with FRI, num_trace_randomizers, the domains (stark.rs:1885-2089, fri.rs:797-920)
    stir = None
    if ldt_choice == "stir":
        # the STIR parameter derivation (f64 formulas pinned by the reference's own tables, tests/test_stir_parameters.py)
        # is the one restatement the oracle shares with the product
        from triton_vm_amd.low_degree_test import stark_stir

        stir = stark_stir(padded_height, security_level=security_level, log2_ldt_expansion_factor=log2_expansion)
        checks = stir.num_first_round_queries()
    else:
        checks = fri_num_collinearity_checks(security_level, log2_expansion)
    h = checks + 4 * 3 * 2 + 1
    rtl = 1 << (max(padded_height + h, 2 * h + 1, (h + 1) * 5) - 1).bit_length()
    trace_len = rtl // 2
    log2_hdb = log2_padded_height
    while True:
        log2_hdb += 1
        ldt_len = 1 << (log2_hdb + log2_expansion)
        if ldt_len >= rtl << log2_expansion:
            break
    g = orc.lib().orc_bfe_generator()
    if stir is not None:
        ldt_len = stir.initial_domain.length
    ldt = orc.domain_of_length(ldt_len, offset=g)
    fri_max_degree = (ldt_len >> log2_expansion) - 1
    fri_rounds = max(0, (fri_max_degree + 1).bit_length() - 1 - (checks.bit_length() - 1 + 1))

    main_root = view.dequeue("MerkleRoot")
    sampled = view.sample_scalars(59)
    challenges = T.derive_challenges([values(c) for c in sampled], values(claim.program_digest), values(claim.input), values(claim.output))
    ch = orc.to_mont(np.array(challenges, dtype=object))
    aux_root = view.dequeue("MerkleRoot")
    quot_weights = _powers(view.sample_scalars(1)[0], NUM_CONSTRAINTS)
    quot_root = view.dequeue("MerkleRoot")

    trace_generator = orc.domain_of_length(trace_len).generator
    alpha = view.sample_scalars(1)[0]
    alpha_next = np.array([orc.lib().orc_bfe_mul(int(c), trace_generator) for c in alpha], np.uint64)
    zeta = orc.bfe(3)
    alpha_zeta = np.array([orc.lib().orc_bfe_mul(int(c), zeta) for c in alpha], np.uint64)
    a4, za4 = orc.xfe_pow(alpha, 4), orc.xfe_pow(alpha_zeta, 4)
    row = lambda name, n: np.asarray(view.dequeue(name), np.uint64).reshape(n, 3)
    main_cur, aux_cur = row("OutOfDomainMainRow", NUM_MAIN), row("OutOfDomainAuxRow", NUM_AUX)
    main_next, aux_next = row("OutOfDomainMainRow", NUM_MAIN), row("OutOfDomainAuxRow", NUM_AUX)
    seg_p, seg_r = row("OutOfDomainQuotientSegments", 4), row("OutOfDomainQuotientSegments", 4)

    # the out-of-domain quotient value from the AIR (stark.rs:1466-1523)
    constraints = orc.air_constraint_values(main_cur, main_next, aux_cur, aux_next, ch)
    one = _lift(orc.bfe(1))
    cons_z_inv = orc.xfe_inv(orc.xfe_sub(orc.xfe_pow(alpha, trace_len), one))
    except_last = orc.xfe_sub(alpha, _lift(orc.lib().orc_bfe_inv(trace_generator)))
    z_inv = {"init": orc.xfe_inv(orc.xfe_sub(alpha, one)), "cons": cons_z_inv, "tran": orc.xfe_mul(except_last, cons_z_inv),
             "term": orc.xfe_inv(except_last)}
    ood_quotient = _xsum(orc.xfe_mul(quot_weights[i], orc.xfe_mul(constraints[i], z_inv[name]))
                         for name, a, b in SECTIONS for i in range(a, b))
    derandomized = orc.xfe_add(_xsum(orc.xfe_mul(orc.xfe_pow(alpha, i), seg_p[i]) for i in range(4)),
                               _xsum(orc.xfe_mul(orc.xfe_pow(alpha_zeta, i), seg_r[i]) for i in range(4)))
    if not (ood_quotient == derandomized).all():
        raise VerificationError("OutOfDomainQuotientValueMismatch")

    # combination weights and the out-of-domain sums (stark.rs:1541-1575)
    iw = view.sample_scalars(3)
    w_ma, w_q, w_d = _powers(iw[0], NUM_MAIN + NUM_AUX), _powers(iw[1], 5), _powers(iw[2], 4)

    def linearly_sum(main_row, aux_row):
        main_row = np.asarray(main_row, np.uint64)
        lifted = main_row if main_row.ndim == 2 else np.stack([main_row, np.zeros_like(main_row), np.zeros_like(main_row)], 1)
        return _xsum([orc.xfe_mul(w_ma[i], lifted[i]) for i in range(NUM_MAIN)]
                     + [orc.xfe_mul(w_ma[NUM_MAIN + i], aux_row[i]) for i in range(NUM_AUX)])

    ood_ma_cur, ood_ma_next = linearly_sum(main_cur, aux_cur), linearly_sum(main_next, aux_next)
    ood_p = _xsum(orc.xfe_mul(seg_p[k], w_q[k]) for k in range(4))
    ood_r = _xsum(orc.xfe_mul(seg_r[k], w_q[k + 1]) for k in range(4))

    # the low-degree test (stark.rs:1577-1590)
    postscript = {}
    if stir is not None:
        indices = lv.stir_verify(view, stir, postscript)
    else:
        indices = lv.fri_verify(view, ldt, fri_rounds, checks, fri_max_degree >> fri_rounds, postscript)
    revealed = postscript["partial_first_codeword"]
    if len(indices) != checks or len(revealed) != checks:
        raise VerificationError("IncorrectNumberOfRowIndices")

    # the revealed rows against their roots (stark.rs:1592-1672)
    def rows_of(name, width, root, error):
        rows = np.asarray(view.dequeue(name), np.uint64).reshape(-1, width)
        if len(rows) != checks:
            raise VerificationError(f"IncorrectNumberOf{name}")
        auth = np.asarray(view.dequeue("AuthenticationStructure"), np.uint64).reshape(-1, 5)
        try:
            lv.verify_inclusion(root, ldt_len, indices, orc.hash_rows(rows), auth)
        except VerificationError:
            raise VerificationError(error)
        return rows

    main_rows = rows_of("MasterMainTableRows", NUM_MAIN, main_root, "MainCodewordAuthenticationFailure")
    aux_rows = rows_of("MasterAuxTableRows", NUM_AUX * 3, aux_root, "AuxiliaryCodewordAuthenticationFailure").reshape(-1, NUM_AUX, 3)
    quot_rows = rows_of("QuotientSegmentsElements", 15, quot_root, "QuotientCodewordAuthenticationFailure").reshape(-1, 5, 3)

    # the combination codeword at the revealed rows (stark.rs:1674-1755)
    def deep_update(x, value, ood_point, ood_value):
        return orc.xfe_mul(orc.xfe_sub(value, ood_value), orc.xfe_inv(orc.xfe_sub(_lift(x), ood_point)))

    for i, main_row, aux_row, seg, want in zip(indices, main_rows, aux_rows, quot_rows, revealed):
        x = lv.domain_value(ldt, i)
        ma = linearly_sum(main_row, aux_row)
        shared = _xsum(orc.xfe_mul(seg[k], w_q[k]) for k in (1, 2, 3))
        p = orc.xfe_add(orc.xfe_mul(w_q[0], seg[0]), shared)
        r = orc.xfe_add(orc.xfe_mul(w_q[4], seg[4]), shared)
        parts = [deep_update(x, ma, alpha, ood_ma_cur), deep_update(x, ma, alpha_next, ood_ma_next),
                 deep_update(x, p, a4, ood_p), deep_update(x, r, za4, ood_r)]
        if not (_xsum(orc.xfe_mul(w_d[k], parts[k]) for k in range(4)) == want).all():
            raise VerificationError("CombinationCodewordMismatch")
    if view.pending:
        raise VerificationError("SuperfluousProofItems")
    return indices
