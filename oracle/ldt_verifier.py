"""TEST INFRASTRUCTURE ONLY -- CPU restatements of the verifier halves of the two low-degree tests
(/root/reference/triton-vm/src/low_degree_test/fri.rs:368-700 and stir.rs:995-1340), so that the device provers can be
tested the way the reference tests its own: prove, then verify; honest codewords are accepted, high-degree ones and
corrupted transcripts are rejected.  They read the proof stream of triton_vm_amd/proof_stream.py (ProofStream
.verifier_view()); authentication structures are twenty-first's (MerkleTree::authentication_structure).
Pure Python over the C oracle's arithmetic: small cases only.  The product never imports this file.
"""
import numpy as np

from . import oracle as orc
from . import stir_oracle as so


class VerificationError(Exception):
    pass


def _sibling_index_rule(n_leaves, leaf_indices):
    """[twenty-first MerkleTree::authentication_structure] the node indices of an authentication structure, in its
    order: the siblings along the paths that are not themselves on a path, descending"""
    k = np.unique(np.asarray(leaf_indices, dtype=np.uint64) + np.uint64(n_leaves))
    needed, computable = [], []
    while k.size and k[0] > 1:
        computable.append(k)
        needed.append(k ^ np.uint64(1))
        k = np.unique(k >> np.uint64(1))
    if not needed:
        return np.zeros(0, np.uint64)
    return np.setdiff1d(np.concatenate(needed), np.concatenate(computable))[::-1]


def verify_inclusion(root, n_leaves, leaf_indices, leaf_digests, auth_nodes):
    """MerkleTreeInclusionProof::verify: recompute the root from the opened leaves and the sibling nodes"""
    known = {}
    for i, d in zip(leaf_indices, leaf_digests):
        key = int(i) + n_leaves
        if key in known and not (known[key] == d).all():
            raise VerificationError("BadMerkleAuthenticationPath: two different leaves at one index")
        known[key] = np.asarray(d, np.uint64)
    sent = _sibling_index_rule(n_leaves, leaf_indices)
    if len(sent) != len(auth_nodes):
        raise VerificationError("BadMerkleAuthenticationPath: wrong number of authentication nodes")
    provided = {int(i): np.asarray(d, np.uint64) for i, d in zip(sent, auth_nodes)}
    level = sorted(known)
    while level and level[0] > 1:
        parents = {}
        for k in level:
            sib = k ^ 1
            left, right = (k, sib) if k % 2 == 0 else (sib, k)
            get = lambda j: known[j] if j in known else provided.get(j)  # noqa: E731
            lv, rv = get(left), get(right)
            if lv is None or rv is None:
                raise VerificationError("BadMerkleAuthenticationPath: missing node")
            parents[k >> 1] = orc.hash_pair(lv, rv)
        known.update(parents)
        level = sorted(parents)
    if not (known.get(1) == np.asarray(root, np.uint64)).all():
        raise VerificationError("BadMerkleAuthenticationPath")


def colinear_y(p0, p1, x):
    """Polynomial::get_colinear_y: the line through p0 and p1 at x"""
    (x0, y0), (x1, y1) = p0, p1
    slope = orc.xfe_mul(orc.xfe_sub(y1, y0), orc.xfe_inv(orc.xfe_sub(x1, x0)))
    return orc.xfe_add(y0, orc.xfe_mul(slope, orc.xfe_sub(x, x0)))


def domain_value(d, i):
    return orc.lib().orc_bfe_mul(d.offset, orc.lib().orc_bfe_pow(d.generator, int(i)))


def fri_verify(view, first_domain, num_rounds, num_collinearity_checks, last_round_max_degree, postscript=None):
    """Fri::verify (fri.rs:368-700) over a verifier view of the transcript -> first-round indices; `postscript` (a dict)
    receives the partially revealed first codeword (InitialRoundPostscript::partial_codeword)"""
    rounds = []
    dom = first_domain
    for r in range(num_rounds + 1):
        root = view.dequeue("fri root")
        challenge = view.sample_scalars(1)[0] if r < num_rounds else None
        rounds.append(dict(domain=dom, root=root, challenge=challenge))
        dom = orc.domain_pow(dom, 2)
    last_codeword = np.asarray(view.dequeue("fri last codeword"), np.uint64).reshape(-1, 3)
    last_poly = np.asarray(view.dequeue("fri last polynomial"), np.uint64).reshape(-1, 3)
    last_dom = rounds[-1]["domain"]
    if len(last_codeword) != last_dom.length:
        raise VerificationError("LastCodewordMismatch")
    a0 = view.sample_indices(first_domain.length, num_collinearity_checks)

    def a_indices(r):
        return [i % rounds[r]["domain"].length for i in a0]

    def b_indices(r):
        n = rounds[r]["domain"].length
        return [(i + n // 2) % n for i in a0]

    def receive(r, indices):
        leaves = np.asarray(view.dequeue("fri response"), np.uint64).reshape(-1, 3)
        auth = np.asarray(view.dequeue("fri auth"), np.uint64).reshape(-1, 5)
        if len(leaves) != num_collinearity_checks:
            raise VerificationError("IncorrectNumberOfRevealedLeaves")
        verify_inclusion(rounds[r]["root"], rounds[r]["domain"].length, indices, orc.xfe_to_digest(leaves), auth)
        return leaves

    partial_a = receive(0, a_indices(0))
    if postscript is not None:
        postscript["partial_first_codeword"] = partial_a
    for r in range(num_rounds):
        partial_b = receive(r, b_indices(r))
        d = rounds[r]["domain"]
        nxt = []
        for i, (ia, ib) in enumerate(zip(a_indices(r), b_indices(r))):
            pa = (so.lift(domain_value(d, ia)), partial_a[i])
            pb = (so.lift(domain_value(d, ib)), partial_b[i])
            nxt.append(colinear_y(pa, pb, rounds[r]["challenge"]))
        partial_a = np.array(nxt, np.uint64)
    # the last round: commitment, agreement with the folded values, low degree
    digests = orc.xfe_to_digest(last_codeword)
    if not (orc.merkle_tree(digests)[1] == rounds[-1]["root"]).all():
        raise VerificationError("BadMerkleRootForLastCodeword")
    if not all((last_codeword[i] == partial_a[j]).all() for j, i in enumerate(a_indices(num_rounds))):
        raise VerificationError("LastCodewordMismatch")
    if len(so.poly_trim(list(last_poly))) > last_round_max_degree + 1:
        raise VerificationError("LastRoundPolynomialHasTooHighDegree")
    x = view.sample_scalars(1)[0]
    plain = orc.domain_of_length(last_dom.length)  # the prover interpolates over the offset-free domain (fri.rs:255-268)
    interpolant = orc.coset_interpolate(last_codeword, plain, 3).reshape(-1, 3)
    if not (orc.poly_eval_xfe(last_poly, x) == orc.poly_eval_xfe(interpolant, x)).all():
        raise VerificationError("LastRoundPolynomialEvaluationMismatch")
    return a0


def _stir_queries(view, stir, round_domain, num_queries, root):
    """extract_inclusion_proof + authenticated_queries (stir.rs:1157-1226)"""
    ff = stir.folding_factor
    indices = view.sample_indices(round_domain.length, num_queries)
    leafs = np.asarray(view.dequeue("stir response leafs"), np.uint64).reshape(-1, ff, 3)
    auth = np.asarray(view.dequeue("stir response auth"), np.uint64).reshape(-1, 5)
    folded_domain = orc.domain_pow(round_domain, ff)
    folded = list(dict.fromkeys(i % folded_domain.length for i in indices))
    if len(leafs) != len(folded):
        raise VerificationError("IncorrectNumberOfRevealedLeaves")
    by_index = dict(zip(folded, leafs))
    digests = [orc.hash_varlen(np.ascontiguousarray(by_index[i]).reshape(-1)) for i in folded]
    verify_inclusion(root, folded_domain.length, folded, digests, auth)
    kth_root = orc.lib().orc_bfe_pow(round_domain.generator, folded_domain.length)
    queries = [dict(index=i, point=domain_value(folded_domain, i % folded_domain.length),
                    root=domain_value(round_domain, i % folded_domain.length), kth_root=kth_root,
                    values=by_index[i % folded_domain.length]) for i in indices]
    return indices, queries


def _coset_interpolate_and_evaluate(root, kth_root, values, at):
    """fast_coset_interpolate(root, values).evaluate(at): the polynomial of degree < ff with the given values on
    root * <kth_root>"""
    pts, x = [], root
    for _ in range(len(values)):
        pts.append(so.lift(x))
        x = orc.lib().orc_bfe_mul(x, kth_root)
    coeffs = so.lagrange_interpolate(pts, [np.asarray(v, np.uint64) for v in values])
    return orc.poly_eval_xfe(np.array(coeffs + [so.ZERO] * (len(values) - len(coeffs))), at)


def stir_verify(view, stir, postscript=None):
    """Stir::verify (stir.rs:995-1120) -> first-round indices; `postscript` (a dict) receives the partially revealed
    first codeword (Stir::partial_codeword, stir.rs:1245-1256: of a queried stack, the value at the queried index)"""
    def partial_codeword(round_domain, queries):
        folded_len = round_domain.length // stir.folding_factor
        return np.array([q["values"][q["index"] // folded_len] for q in queries], np.uint64)

    domain = orc.Domain(stir.initial_domain.offset, stir.initial_domain.generator, stir.initial_domain.length)
    previous_root = view.dequeue("stir root")
    previous = None  # (quotient_set, quotient_answers, degree_correction_randomness)
    first_round_indices = None

    def in_domain_answers(queries, folding_randomness):
        if previous is None:
            return [_coset_interpolate_and_evaluate(q["root"], q["kth_root"], q["values"], folding_randomness) for q in queries]
        qset, qans, rc = previous
        answer_poly = np.array(so.lagrange_interpolate(qset, qans) or [so.ZERO])
        zerofier = np.array(so.zerofier(qset))
        e = len(qset) + 1
        out = []
        for q in queries:
            evaluations, x = [], q["root"]
            for value in q["values"]:
                quotient = orc.xfe_mul(orc.xfe_sub(value, orc.poly_eval_xfe(answer_poly, so.lift(x))),
                                       orc.xfe_inv(orc.poly_eval_xfe(zerofier, so.lift(x))))
                common = orc.xfe_mul(so.lift(x), rc)
                if (common == so.one()).all():
                    factor = so.lift(orc.bfe(e))
                else:
                    factor = orc.xfe_mul(orc.xfe_sub(so.one(), orc.xfe_pow(common, e)),
                                         orc.xfe_inv(orc.xfe_sub(so.one(), common)))
                evaluations.append(orc.xfe_mul(factor, quotient))
                x = orc.lib().orc_bfe_mul(x, q["kth_root"])
            out.append(_coset_interpolate_and_evaluate(q["root"], q["kth_root"], evaluations, folding_randomness))
        return out

    for in_domain, out_of_domain in stir.round_queries:
        folding_randomness = view.sample_scalars(1)[0]
        current_root = view.dequeue("stir root")
        ood_queries = view.sample_scalars(out_of_domain)
        ood_answers = np.asarray(view.dequeue("stir ood values"), np.uint64).reshape(-1, 3)
        indices, queries = _stir_queries(view, stir, domain, in_domain, previous_root)
        if postscript is not None and "partial_first_codeword" not in postscript:
            postscript["partial_first_codeword"] = partial_codeword(domain, queries)
        answers = in_domain_answers(queries, folding_randomness)
        qset, qans, seen = [], [], set()
        for point, answer in list(zip([so.lift(q["point"]) for q in queries], answers)) + list(zip(ood_queries, ood_answers)):
            key = tuple(int(c) for c in point)
            if key not in seen:
                seen.add(key)
                qset.append(np.asarray(point, np.uint64))
                qans.append(np.asarray(answer, np.uint64))
        previous = (qset, qans, view.sample_scalars(1)[0])
        if first_round_indices is None:
            first_round_indices = indices
        nxt = orc.domain_pow(domain, 2)
        domain = orc.Domain(orc.lib().orc_bfe_mul(nxt.offset, domain.offset), nxt.generator, nxt.length)
        previous_root = current_root

    folding_randomness = view.sample_scalars(1)[0]
    poly = np.asarray(view.dequeue("stir final polynomial"), np.uint64).reshape(-1, 3)
    if max(len(so.poly_trim(list(poly))) - 1, 0) > stir.final_degree:
        raise VerificationError("LastRoundPolynomialHasTooHighDegree")
    indices, queries = _stir_queries(view, stir, domain, stir.final_num_in_domain_queries, previous_root)
    if postscript is not None and "partial_first_codeword" not in postscript:
        postscript["partial_first_codeword"] = partial_codeword(domain, queries)
    for q, answer in zip(queries, in_domain_answers(queries, folding_randomness)):
        if not (orc.poly_eval_xfe(poly, so.lift(q["point"])) == answer).all():
            raise VerificationError("LastRoundPolynomialEvaluationMismatch")
    return first_round_indices if first_round_indices is not None else indices
