"""Host mirror of the verifier: its batch work over the revealed rows (/root/reference/triton-vm/src/stark.rs:1598-1601,
1620-1660, 1678-1755) over the C ABI (csrc/verify.hip: `row_digests`, `deep_values`), and `Verifier.verify` -- the whole of
Verifier::verify (stark.rs:1388-1763) with FRI -- which sequences them with the Fiat-Shamir schedule and the decisions."""
import ctypes as C

import numpy as np


def _h(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def row_digests(ctx, rows):
    """Tip5::hash_varlen of every revealed row (leaf digests, stark.rs:1598-1601): rows [n][words] -> [n][5]"""
    rows = _h(rows)
    rows = rows.reshape(rows.shape[0], -1)
    out = np.empty((rows.shape[0], 5), np.uint64)
    ctx._check(ctx.lib.tvm_verifier_row_digests(ctx.handle, rows.ctypes.data, rows.shape[0], rows.shape[1], out.ctypes.data),
               "tvm_verifier_row_digests")
    return out


def deep_values(ctx, main_rows, aux_rows, quotient_rows, row_indices, ldt_domain, weights_main_aux, weights_quot, weights_deep,
                ood_points, ood_values):
    """The value the combination codeword must have at each revealed index (stark.rs:1678-1755): [n][3].
    ood_points / ood_values in the order current row, next row, alpha^4, (zeta * alpha)^4."""
    main_rows, aux_rows, quotient_rows = _h(main_rows), _h(aux_rows), _h(quotient_rows)
    idx = _h(row_indices)
    n = idx.size
    if main_rows.size != n * 379 or aux_rows.size != n * 273 or quotient_rows.size != n * 15:
        raise ValueError("revealed rows must be [n][379], [n][91][3] and [n][5][3]")
    wma, wq, wd = _h(weights_main_aux).reshape(470, 3), _h(weights_quot).reshape(5, 3), _h(weights_deep).reshape(4, 3)
    pts, vals = _h(ood_points).reshape(4, 3), _h(ood_values).reshape(4, 3)
    out = np.empty((n, 3), np.uint64)
    ctx._check(ctx.lib.tvm_verifier_deep_values(ctx.handle, main_rows.ctypes.data, aux_rows.ctypes.data, quotient_rows.ctypes.data,
                                                idx.ctypes.data, n, ldt_domain.c(), wma.ctypes.data, wq.ctypes.data, wd.ctypes.data,
                                                pts.ctypes.data, vals.ctypes.data, out.ctypes.data), "tvm_verifier_deep_values")
    return out


# ------------------------------------------------------------------------------------------------ Verifier::verify
class VerificationError(Exception):
    """error.rs: VerificationError / LdtVerificationError, by variant name"""


def _xfe(lib):
    """XFieldElement arithmetic on Montgomery words for the handful of scalar steps of the verifier"""
    from . import field

    class X:
        @staticmethod
        def mul(a, b):
            o = np.zeros(3, np.uint64)
            a, b = _h(a), _h(b)
            lib.tvm_host_xfe_mul(a.ctypes.data, b.ctypes.data, o.ctypes.data)
            return o

        @staticmethod
        def inv(a):
            o = np.zeros(3, np.uint64)
            a = _h(a)
            lib.tvm_host_xfe_inv(a.ctypes.data, o.ctypes.data)
            return o

        @staticmethod
        def add(a, b):
            return np.array([(int(x) + int(y)) % field.P for x, y in zip(a, b)], np.uint64)

        @staticmethod
        def sub(a, b):
            return np.array([(int(x) - int(y)) % field.P for x, y in zip(a, b)], np.uint64)

        @staticmethod
        def lift(b):
            return np.array([int(b), 0, 0], np.uint64)

        @staticmethod
        def powers(x, n, first=0):
            o = np.zeros((n, 3), np.uint64)
            x = _h(x)
            lib.tvm_host_xfe_powers(x.ctypes.data, first, n, o.ctypes.data)
            return o

        @staticmethod
        def sum(terms):
            acc = np.zeros(3, np.uint64)
            for t in terms:
                acc = X.add(acc, t)
            return acc

    return X


def _trimmed(polynomial):
    """coefficients without trailing zeros (a decoded Polynomial has none; a prover's own stream may)"""
    c = _h(polynomial).reshape(-1, 3)
    n = len(c)
    while n and not c[n - 1].any():
        n -= 1
    return c[:n]


def hash_pair(lib, left, right):
    """Tip5::hash_pair on the host: the fixed-length domain (capacity of ones)"""
    from . import field

    state = np.concatenate([_h(left).reshape(5), _h(right).reshape(5), np.full(6, field.ONE, np.uint64)])
    lib.tvm_host_tip5_permutation(state.ctypes.data)
    return state[:5].copy()


def verify_inclusion(lib, root, n_leaves, leaf_indices, leaf_digests, authentication_structure, error):
    """[twenty-first MerkleTreeInclusionProof::verify, restated] recompute the root from the revealed leaves and the
    authentication structure (the nodes of stark.auth_node_indices, in that order)"""
    from .stark import auth_node_indices

    known = {}
    for i, d in zip(leaf_indices, leaf_digests):
        key = int(i) + n_leaves
        d = _h(d)
        if d.shape != (5,) or (key in known and not np.array_equal(known[key], d)):
            raise VerificationError(error)
        known[key] = d
    sent = auth_node_indices(n_leaves, leaf_indices)
    auth = _h(authentication_structure).reshape(-1, 5)
    if len(sent) != len(auth):
        raise VerificationError(error)
    known.update({int(i): d for i, d in zip(sent, auth) if int(i) not in known})
    level = sorted(k for k in known if k >= n_leaves)
    while level and level[0] > 1:
        parents = {}
        for k in level:
            left, right = known.get(k & ~1), known.get(k | 1)
            if left is None or right is None:
                raise VerificationError(error)
            parents[k >> 1] = hash_pair(lib, left, right)
        known.update(parents)
        level = sorted(parents)
    root = _h(root)
    if root.shape != (5,) or not np.array_equal(known.get(1), root):   # (never an elementwise broadcast against a short root)
        raise VerificationError(error)


class Verifier:
    """Verifier::verify (/root/reference/triton-vm/src/stark.rs:1388-1763) with FRI as the low-degree test (fri.rs:368-700):
    the host sequences the Fiat-Shamir schedule and the decisions; the row hashing and the per-row combination values run
    on the device (tvm_verifier_row_digests, tvm_verifier_deep_values), the AIR at the out-of-domain rows through
    tvm_host_air_constraints.  ldt: "fri" or "stir" (stir.rs:995-1340) -- Stark::ldt picks by padded height
    (stark.rs:1944-1951: STIR from 2^16 on); None (the default, like Stark::default()) applies that rule."""

    def __init__(self, ctx, security_level=160, log2_expansion=2, ldt=None):
        self.ctx, self.security_level, self.log2_expansion, self.ldt = ctx, security_level, log2_expansion, ldt

    def verify(self, claim, proof_words):
        """raises VerificationError / ProofDecodingError; returns the revealed row indices on acceptance"""
        import math

        from . import field
        from .arithmetic_domain import ArithmeticDomain
        from .low_degree_test import ReedSolomonCode
        from .proof_stream import ProofStream
        from .prover import NUM_AUX, NUM_CONSTRAINTS, NUM_MAIN, NUM_SAMPLED_CHALLENGES, StarkParameters, derive_challenges
        from .stark import ZETA

        ctx, lib = self.ctx, self.ctx.lib
        X = _xfe(lib)
        view = ProofStream.from_proof(lib, proof_words).verifier_view()

        def dequeue(variant):
            try:
                return view.dequeue(variant)
            except ValueError as e:
                raise VerificationError(f"ProofStreamError: {e}")

        view.alter_fiat_shamir_state_with(claim.encode())
        log2_padded_height = field.from_mont(int(dequeue("Log2PaddedHeight")[0]))
        if log2_padded_height >= 32:
            raise VerificationError("Log2PaddedHeightTooLarge")
        ldt = self.ldt or ("fri" if log2_padded_height < 16 else "stir")
        if ldt == "stir":
            from .low_degree_test import stark_stir

            stir = stark_stir(1 << log2_padded_height, security_level=self.security_level, log2_ldt_expansion_factor=self.log2_expansion)
            checks = stir.num_first_round_queries()
            p = StarkParameters(log2_padded_height, num_trace_randomizers=stir.num_trace_randomizers(), log2_expansion=self.log2_expansion)
            p.ldt = stir.initial_domain
        else:
            checks = math.ceil(-self.security_level / math.log2(1.0 - ReedSolomonCode(self.log2_expansion).proximity_parameter()))
            p = StarkParameters(log2_padded_height, num_trace_randomizers=checks + 4 * 3 * 2 + 1, num_collinearity_checks=checks,
                                log2_expansion=self.log2_expansion)
        L = p.ldt.length

        # Fiat-Shamir 1 (stark.rs:1418-1437)
        main_root = dequeue("MerkleRoot")
        challenges = derive_challenges(lib, view.sample_scalars(NUM_SAMPLED_CHALLENGES), claim)
        aux_root = dequeue("MerkleRoot")
        quotient_weights = X.powers(view.sample_scalars(1)[0], NUM_CONSTRAINTS)
        quot_root = dequeue("MerkleRoot")

        # the out-of-domain rows and the quotient value they imply (stark.rs:1439-1539)
        alpha = view.sample_scalars(1)[0]
        scale = lambda x, b: np.array([field.mont_mul(int(c), b) for c in x], np.uint64)
        alpha_next, alpha_zeta = scale(alpha, p.trace.generator), scale(alpha, ZETA)
        a4, za4 = X.powers(alpha, 1, 4)[0], X.powers(alpha_zeta, 1, 4)[0]
        row = lambda variant, n: _h(dequeue(variant)).reshape(n, 3)
        main_cur, aux_cur = row("OutOfDomainMainRow", NUM_MAIN), row("OutOfDomainAuxRow", NUM_AUX)
        main_next, aux_next = row("OutOfDomainMainRow", NUM_MAIN), row("OutOfDomainAuxRow", NUM_AUX)
        seg_p, seg_r = row("OutOfDomainQuotientSegments", 4), row("OutOfDomainQuotientSegments", 4)
        constraints = np.zeros((NUM_CONSTRAINTS, 3), np.uint64)
        ctx._check(lib.tvm_host_air_constraints(main_cur.ctypes.data, aux_cur.ctypes.data, main_next.ctypes.data, aux_next.ctypes.data,
                                                _h(challenges).ctypes.data, constraints.ctypes.data), "tvm_host_air_constraints")
        one = X.lift(field.ONE)
        consistency_inv = X.inv(X.sub(X.powers(alpha, 1, p.trace.length)[0], one))
        except_last = X.sub(alpha, X.lift(field.mont_inv(p.trace.generator)))
        zerofier_inverse = [(81, X.inv(X.sub(alpha, one))), (97, consistency_inv), (403, X.mul(except_last, consistency_inv)),
                            (23, X.inv(except_last))]          # initial, consistency, transition, terminal (stark.rs:1493-1499)
        summands, k = [], 0
        for count, z_inv in zerofier_inverse:
            summands += [X.mul(quotient_weights[i], X.mul(constraints[i], z_inv)) for i in range(k, k + count)]
            k += count
        ood_quotient = X.sum(summands)
        derandomized = X.add(X.sum(X.mul(w, x) for w, x in zip(X.powers(alpha, 4), seg_p)),
                             X.sum(X.mul(w, x) for w, x in zip(X.powers(alpha_zeta, 4), seg_r)))
        if not (ood_quotient == derandomized).all():
            raise VerificationError("OutOfDomainQuotientValueMismatch")

        # Fiat-Shamir 2 and the out-of-domain sums (stark.rs:1541-1575)
        iw = view.sample_scalars(3)
        w_ma, w_q, w_d = X.powers(iw[0], NUM_MAIN + NUM_AUX), X.powers(iw[1], 5), X.powers(iw[2], 4)
        linear_sum = lambda m, a: X.sum([X.mul(w_ma[i], m[i]) for i in range(NUM_MAIN)]
                                        + [X.mul(w_ma[NUM_MAIN + i], a[i]) for i in range(NUM_AUX)])
        ood_values = [linear_sum(main_cur, aux_cur), linear_sum(main_next, aux_next),
                      X.sum(X.mul(seg_p[i], w_q[i]) for i in range(4)), X.sum(X.mul(seg_r[i], w_q[i + 1]) for i in range(4))]

        # the low-degree test (stark.rs:1577-1590)
        indices, revealed = self._stir_verify(view, dequeue, stir, X) if ldt == "stir" else self._fri_verify(view, dequeue, p, X)
        if len(indices) != checks or len(revealed) != checks:
            raise VerificationError("IncorrectNumberOfRowIndices")

        # the revealed rows against their roots, hashed on the device (stark.rs:1592-1672)
        def rows_of(variant, width, root, error):
            rows = _h(dequeue(variant)).reshape(-1, width)
            if len(rows) != checks:
                raise VerificationError(f"IncorrectNumberOf{variant}")
            auth = dequeue("AuthenticationStructure")
            verify_inclusion(lib, root, L, indices, row_digests(ctx, rows), auth, error)
            return rows

        main_rows = rows_of("MasterMainTableRows", NUM_MAIN, main_root, "MainCodewordAuthenticationFailure")
        aux_rows = rows_of("MasterAuxTableRows", NUM_AUX * 3, aux_root, "AuxiliaryCodewordAuthenticationFailure")
        quot_rows = rows_of("QuotientSegmentsElements", 15, quot_root, "QuotientCodewordAuthenticationFailure")

        # the combination codeword at the revealed rows, on the device (stark.rs:1674-1755)
        want = deep_values(ctx, main_rows, aux_rows, quot_rows, indices, p.ldt, w_ma, w_q, w_d, [alpha, alpha_next, a4, za4], ood_values)
        if not (want == revealed).all():
            raise VerificationError("CombinationCodewordMismatch")
        if view.pending:
            raise VerificationError("SuperfluousProofItems")
        return indices

    def _fri_verify(self, view, dequeue, p, X):
        """Fri::verify (fri.rs:368-700) -> (first-round indices, the partially revealed first codeword)"""
        from . import field, stark
        from .arithmetic_domain import ArithmeticDomain

        ctx, lib = self.ctx, self.ctx.lib
        checks, num_rounds = p.num_collinearity_checks, p.fri_rounds
        rounds, dom = [], p.ldt
        for r in range(num_rounds + 1):
            root = dequeue("MerkleRoot")
            rounds.append((dom, root, view.sample_scalars(1)[0] if r < num_rounds else None))
            dom = dom.pow(2)
        last_domain = rounds[-1][0]
        last_codeword = _h(dequeue("FriCodeword")).reshape(-1, 3)
        last_polynomial = _trimmed(dequeue("Polynomial"))
        if len(last_codeword) != last_domain.length:
            raise VerificationError("LastCodewordMismatch")
        a0 = view.sample_indices(p.ldt.length, checks)
        digest_of = lambda leaves: np.concatenate([_h(leaves).reshape(-1, 3), np.zeros((len(leaves), 2), np.uint64)], axis=1)

        def receive(r, indices):
            leaves = _h(dequeue("fri response")).reshape(-1, 3)
            auth = dequeue("fri auth")
            if len(leaves) != checks:
                raise VerificationError("IncorrectNumberOfRevealedLeaves")
            verify_inclusion(lib, rounds[r][1], rounds[r][0].length, indices, digest_of(leaves), auth, "BadMerkleAuthenticationPath")
            return leaves

        first = partial_a = receive(0, a0)
        for r in range(num_rounds):
            d, _, challenge = rounds[r]
            ia = [i % d.length for i in a0]
            ib = [(i + d.length // 2) % d.length for i in a0]
            partial_b = receive(r, ib)
            folded = []
            for j in range(checks):      # Polynomial::get_colinear_y: the line through the two points, at the challenge
                xa, xb = X.lift(d.value(ia[j])), X.lift(d.value(ib[j]))
                slope = X.mul(X.sub(partial_b[j], partial_a[j]), X.inv(X.sub(xb, xa)))
                folded.append(X.add(partial_a[j], X.mul(slope, X.sub(challenge, xa))))
            partial_a = np.array(folded, np.uint64)
        # the last round: commitment, agreement with the folded values, low degree (fri.rs:560-640)
        nodes = {last_domain.length + i: d for i, d in enumerate(digest_of(last_codeword))}
        for k in range(last_domain.length - 1, 0, -1):
            nodes[k] = hash_pair(lib, nodes[2 * k], nodes[2 * k + 1])
        if not (nodes[1] == _h(rounds[-1][1])).all():
            raise VerificationError("BadMerkleRootForLastCodeword")
        if not all((last_codeword[i % last_domain.length] == partial_a[j]).all() for j, i in enumerate(a0)):
            raise VerificationError("LastCodewordMismatch")
        max_degree = (p.ldt.length >> p.log2_expansion) - 1 >> num_rounds
        if len(last_polynomial) > max_degree + 1:
            raise VerificationError("LastRoundPolynomialHasTooHighDegree")
        x = view.sample_scalars(1)[0]
        d_codeword = ctx.to_device(last_codeword)
        interpolant = ArithmeticDomain.of_length(last_domain.length).interpolate(ctx, d_codeword, 3)
        at_x = stark.evaluate_at_points(ctx, interpolant, last_domain.length, [x])[0]
        claimed = np.zeros(3, np.uint64)
        for c in last_polynomial[::-1]:
            claimed = X.add(X.mul(claimed, x), c)
        if not (claimed == at_x).all():
            raise VerificationError("LastRoundPolynomialEvaluationMismatch")
        return a0, first

    def _stir_verify(self, view, dequeue, stir, X):
        """Stir::verify (stir.rs:995-1340) -> (first-round indices, the partially revealed first codeword)"""
        from . import field
        from .low_degree_test import Stir

        ctx, lib, ff = self.ctx, self.ctx.lib, stir.folding_factor

        def poly_eval(coefficients, points, zerofier=False):
            c, pts = _h(coefficients).reshape(-1, 3), _h(points).reshape(-1, 3)
            out = np.zeros((len(pts), 3), np.uint64)
            lib.tvm_host_xfe_poly_eval(c.ctypes.data, len(c), pts.ctypes.data, len(pts), 1 if zerofier else 0, out.ctypes.data)
            return out

        def interpolate(points, values):
            pts, vals = _h(points).reshape(-1, 3), _h(values).reshape(-1, 3)
            out = np.zeros((len(pts), 3), np.uint64)
            if lib.tvm_host_xfe_interpolate(pts.ctypes.data, vals.ctypes.data, len(pts), out.ctypes.data):
                raise VerificationError("repeated point in an interpolation")
            return out

        def queries(domain, num_queries, root):
            """extract_inclusion_proof + authenticated_queries (stir.rs:1157-1226)"""
            indices = view.sample_indices(domain.length, num_queries)
            leafs = _h(dequeue("stir response leafs"))
            auth = dequeue("stir response auth")
            folded_len = domain.length // ff
            folded = list(dict.fromkeys(i % folded_len for i in indices))
            if leafs.ndim != 3 or leafs.shape[1:] != (ff, 3) or len(leafs) != len(folded):
                raise VerificationError("IncorrectNumberOfRevealedLeaves")
            verify_inclusion(lib, root, folded_len, folded, row_digests(ctx, leafs.reshape(len(folded), ff * 3)), auth,
                             "BadMerkleAuthenticationPath")
            by_index = dict(zip(folded, leafs))
            folded_domain = domain.pow(ff)
            kth_root = field.mont_pow(domain.generator, folded_len)
            return indices, [dict(index=i, point=folded_domain.value(i % folded_len), root=domain.value(i % folded_len), kth_root=kth_root,
                                  values=by_index[i % folded_len]) for i in indices]

        def fold_at(query, values, randomness):
            """fast_coset_interpolate(root, values).evaluate(randomness): degree < ff through root * <kth_root>"""
            pts, x = [], query["root"]
            for _ in range(ff):
                pts.append(X.lift(x))
                x = field.mont_mul(x, query["kth_root"])
            return poly_eval(interpolate(pts, values), [randomness])[0]

        def partial_codeword(domain, qs):
            return np.array([q["values"][q["index"] // (domain.length // ff)] for q in qs], np.uint64)

        def in_domain_answers(qs, folding_randomness, previous):
            if previous is None:        # initial_in_domain_answers (stir.rs:1259-1268)
                return [fold_at(q, q["values"], folding_randomness) for q in qs]
            quotient_set, quotient_answers, rc = previous   # subsequent_in_domain_answers (stir.rs:1270-1340)
            answer_poly = interpolate(quotient_set, quotient_answers)
            e = len(quotient_set) + 1
            one = X.lift(field.ONE)
            out = []
            for q in qs:
                xs, x = [], q["root"]
                for _ in range(ff):
                    xs.append(X.lift(x))
                    x = field.mont_mul(x, q["kth_root"])
                answers, zerofiers = poly_eval(answer_poly, xs), poly_eval(quotient_set, xs, zerofier=True)
                evaluations = []
                for j in range(ff):
                    quotient = X.mul(X.sub(q["values"][j], answers[j]), X.inv(zerofiers[j]))
                    common = X.mul(xs[j], rc)
                    if (common == one).all():
                        factor = X.lift(field.to_mont(e))
                    else:
                        factor = X.mul(X.sub(one, X.powers(common, 1, e)[0]), X.inv(X.sub(one, common)))
                    evaluations.append(X.mul(factor, quotient))
                out.append(fold_at(q, evaluations, folding_randomness))
            return out

        domain = stir.initial_domain
        previous_root = dequeue("MerkleRoot")
        previous = first_indices = first_codeword = None
        for in_domain, out_of_domain in stir.round_queries:
            folding_randomness = view.sample_scalars(1)[0]
            current_root = dequeue("MerkleRoot")
            ood_queries = view.sample_scalars(out_of_domain)
            ood_answers = _h(dequeue("StirOutOfDomainValues")).reshape(-1, 3)
            if len(ood_answers) != out_of_domain:
                raise VerificationError("IncorrectNumberOfOutOfDomainValues")
            indices, qs = queries(domain, in_domain, previous_root)
            if first_indices is None:
                first_indices, first_codeword = indices, partial_codeword(domain, qs)
            answers = in_domain_answers(qs, folding_randomness, previous)
            quotient_set, quotient_answers, seen = [], [], set()   # queried indices repeat; interpolation points must not
            for point, answer in list(zip([X.lift(q["point"]) for q in qs], answers)) + list(zip(ood_queries, ood_answers)):
                key = tuple(int(c) for c in point)
                if key not in seen:
                    seen.add(key)
                    quotient_set.append(_h(point))
                    quotient_answers.append(_h(answer))
            previous = (np.array(quotient_set, np.uint64), np.array(quotient_answers, np.uint64), view.sample_scalars(1)[0])
            domain, previous_root = Stir.next_round_domain(domain), current_root
        folding_randomness = view.sample_scalars(1)[0]
        final = _trimmed(dequeue("Polynomial"))
        if max(len(final) - 1, 0) > stir.final_degree:
            raise VerificationError("LastRoundPolynomialHasTooHighDegree")
        indices, qs = queries(domain, stir.final_num_in_domain_queries, previous_root)
        if first_indices is None:
            first_indices, first_codeword = indices, partial_codeword(domain, qs)
        want = poly_eval(final, [X.lift(q["point"]) for q in qs]) if len(final) else np.zeros((len(qs), 3), np.uint64)
        for got, expected in zip(in_domain_answers(qs, folding_randomness, previous), want):
            if not (got == expected).all():
                raise VerificationError("LastRoundPolynomialEvaluationMismatch")
        return first_indices, first_codeword

