"""Host mirror of the reference's low-degree-test set-up (/root/reference/triton-vm/src/low_degree_test/mod.rs and
stir.rs) and of the STIR prover loop (stir.rs:885-993) over the C ABI.

The parameter derivation is f64 arithmetic restated line by line (same operations in the same order, so the same
IEEE-754 results): ReedSolomonCode (mod.rs:93-170), StirParameters::try_into_stir (stir.rs:420-560), the query counts
(stir.rs:597-777) and Stark::stir / num_trace_randomizers (stark.rs:1972-2089).  Pinned by the reference's own
constants: the q-ary entropy table (mod.rs:406-423) and the two worked examples of the over-sampling bound
(stir.rs:739-747: n = 184 and n = 610).
"""
import ctypes as C
import math

import numpy as np

from . import field
from .arithmetic_domain import ArithmeticDomain

LOG2_FIELD_SIZE_F = 191.99999999899228  # ReedSolomonCode::LOG2_FIELD_SIZE (mod.rs:101)
LOG2_FIELD_SIZE = 64 * 3                # StirParameters::LOG2_FIELD_SIZE (stir.rs:404-405)
LOG2_DOMAIN_SHRINKAGE = 1               # stir.rs:412
NUM_QUOTIENT_SEGMENTS, EXTENSION_DEGREE, AIR_FAN_IN = 4, 3, 2


class LdtParameterError(ValueError):
    pass


class ReedSolomonCode:
    """mod.rs:93-170"""

    def __init__(self, log2_expansion_factor, soundness="proven"):
        self.log2_expansion_factor, self.soundness = log2_expansion_factor, soundness

    def rate(self):
        if self.log2_expansion_factor >= 32:
            raise LdtParameterError("TooBigInitialExpansionFactor")
        return 1.0 / float(1 << self.log2_expansion_factor)

    def q_ary_entropy(self):
        rate = self.rate()
        rate_log_rate = rate * -float(self.log2_expansion_factor)
        one_m = (1.0 - rate) * math.log2(1.0 - rate)
        return rate - (rate_log_rate + one_m) / LOG2_FIELD_SIZE_F

    def proximity_margin(self):
        return math.sqrt(self.rate()) if self.soundness == "proven" else self.q_ary_entropy()

    def slackness_factor(self):
        return self.proximity_margin() / 20.0

    def proximity_parameter(self):
        return 1.0 - self.proximity_margin() - self.slackness_factor()

    def log2_list_size(self, log2_poly_degree):
        if self.soundness == "proven":
            list_size = 1.0 / (2.0 * math.sqrt(self.rate()) * self.slackness_factor())
        else:
            list_size = 2.0 ** float(log2_poly_degree) / (self.q_ary_entropy() * self.slackness_factor())
        return math.log2(list_size)


def log2_binomial_coefficient(a, b):
    """stir.rs:779-793 (Kahan-compensated sum of log2 terms)"""
    assert a >= b
    log2_binom = compensation = 0.0
    for i in range(min(b, a - b)):
        summand = math.log2(float(a - i)) - math.log2(float(i + 1))
        corrected = summand - compensation
        nxt = log2_binom + corrected
        compensation = (nxt - log2_binom) - corrected
        log2_binom = nxt
    return log2_binom


class StirParameters:
    """stir.rs:59-110, 403-560"""

    def __init__(self, security_level, log2_initial_expansion_factor, log2_high_degree_bound, log2_folding_factor=2,
                 soundness="proven"):
        self.security_level, self.soundness = security_level, soundness
        self.log2_folding_factor = log2_folding_factor
        self.log2_initial_expansion_factor = log2_initial_expansion_factor
        self.log2_high_degree_bound = log2_high_degree_bound

    def max_degree(self):
        return (1 << self.log2_high_degree_bound) - 1

    def expansion_factor(self):
        return 1 << self.log2_initial_expansion_factor

    def initial_domain(self):
        log2_len = self.log2_high_degree_bound + self.log2_initial_expansion_factor
        if log2_len > 32:
            raise LdtParameterError(f"InitialDomainTooBig({log2_len})")
        return ArithmeticDomain.of_length(1 << log2_len).with_offset(field.generator())

    # -- query counts, stir.rs:597-777 ---------------------------------------------------------------
    def num_unique_in_domain_queries(self, log2_expansion_factor):
        prox = ReedSolomonCode(log2_expansion_factor, self.soundness).proximity_parameter()
        return int(math.ceil(-float(self.security_level) / math.log2(1.0 - prox)))

    def num_total_in_domain_queries(self, log2_domain_len, num_in_domain_queries):
        k_minus_1 = num_in_domain_queries - 1
        assert k_minus_1 >= 0, "internal error: too few queries"
        domain_len = 1 << log2_domain_len
        ell = min(k_minus_1, domain_len // 2)
        log2_u_choose_l = log2_binomial_coefficient(domain_len, ell)
        log2_k_minus_1 = max(math.log2(float(k_minus_1)) if k_minus_1 > 0 else -math.inf, 0.0)
        n = (float(self.security_level) + log2_k_minus_1 + log2_u_choose_l) / (float(log2_domain_len) - log2_k_minus_1)
        return int(math.ceil(n))

    def num_in_domain_queries(self, log2_domain_size, log2_expansion_factor):
        uniques = min(self.num_unique_in_domain_queries(log2_expansion_factor), 1 << log2_domain_size)
        return self.num_total_in_domain_queries(log2_domain_size, uniques)

    def num_ood_queries(self, log2_poly_degree, log2_expansion_factor):
        log2_list = ReedSolomonCode(log2_expansion_factor, self.soundness).log2_list_size(log2_poly_degree)
        n = (float(self.security_level) - 1.0 + 2.0 * log2_list) / float(LOG2_FIELD_SIZE - log2_poly_degree)
        return int(math.ceil(n))

    def try_into_stir(self):
        if self.log2_folding_factor < 2:
            raise LdtParameterError("TooSmallLog2FoldingFactor")
        if self.log2_initial_expansion_factor == 0:
            raise LdtParameterError("TooSmallInitialExpansionFactor")
        if self.log2_high_degree_bound < self.log2_folding_factor:
            raise LdtParameterError("TooLowDegreeOfHighDegreePolynomials")
        folding_factor = 1 << self.log2_folding_factor
        folded_poly_degree = self.max_degree() // folding_factor
        log2_expansion_factor = self.log2_initial_expansion_factor
        initial_domain = self.initial_domain()
        log2_folded_domain_size = initial_domain.length.bit_length() - 1 - self.log2_folding_factor
        round_queries = []
        while folded_poly_degree > folding_factor:
            in_domain = self.num_in_domain_queries(log2_folded_domain_size, log2_expansion_factor)
            log2_next_expansion = log2_expansion_factor + self.log2_folding_factor - LOG2_DOMAIN_SHRINKAGE
            out_of_domain = self.num_ood_queries(folded_poly_degree.bit_length() - 1, log2_next_expansion)
            next_folded_poly_deg = folded_poly_degree // folding_factor
            if in_domain + out_of_domain > next_folded_poly_deg:
                break
            round_queries.append((in_domain, out_of_domain))
            folded_poly_degree = next_folded_poly_deg
            log2_expansion_factor = log2_next_expansion
            log2_folded_domain_size -= LOG2_DOMAIN_SHRINKAGE
        final_in_domain = self.num_in_domain_queries(log2_folded_domain_size, log2_expansion_factor)
        return Stir(initial_domain, folding_factor, round_queries, final_in_domain, folded_poly_degree)


def randomized_trace_len(padded_height, num_trace_randomizers):
    """stark.rs:1885-1896"""
    h = num_trace_randomizers
    total = max(padded_height + h, 2 * h + 1, (h + 1) * 5)
    return 1 << (total - 1).bit_length()


def stark_stir(padded_height, security_level=160, log2_ldt_expansion_factor=2, soundness="proven"):
    """Stark::stir (stark.rs:1972-2032): the smallest STIR instance whose initial domain holds the randomized trace."""
    padded_height = 1 << (padded_height - 1).bit_length()
    params = StirParameters(security_level, log2_ldt_expansion_factor, padded_height.bit_length() - 1, 2, soundness)
    for _ in range(33):
        params.log2_high_degree_bound += 1
        stir = params.try_into_stir()
        h = stir.num_first_round_queries() + NUM_QUOTIENT_SEGMENTS * EXTENSION_DEGREE * AIR_FAN_IN + 1
        if stir.initial_domain.length >= randomized_trace_len(padded_height, h) * params.expansion_factor():
            return stir
    raise LdtParameterError("no suitable STIR parameters found")


class Stir:
    """stir.rs:120-146 and the prover (stir.rs:885-993)."""

    def __init__(self, initial_domain, folding_factor, round_queries, final_num_in_domain_queries, final_degree):
        self.initial_domain, self.folding_factor = initial_domain, folding_factor
        self.round_queries = round_queries                      # [(in_domain, out_of_domain)]
        self.final_num_in_domain_queries, self.final_degree = final_num_in_domain_queries, final_degree

    def num_first_round_queries(self):
        return self.round_queries[0][0] if self.round_queries else self.final_num_in_domain_queries

    def num_trace_randomizers(self):
        """Stark::num_trace_randomizers (stark.rs:2083-2089)"""
        return self.num_first_round_queries() + NUM_QUOTIENT_SEGMENTS * EXTENSION_DEGREE * AIR_FAN_IN + 1

    @staticmethod
    def next_round_domain(domain):
        """stir.rs:1149-1155"""
        nxt = domain.pow(1 << LOG2_DOMAIN_SHRINKAGE)
        return nxt.with_offset(field.mont_mul(nxt.offset, domain.offset))

    # -- the prover (stir.rs:885-993) -------------------------------------------------------------------
    def prove(self, ctx, d_codeword, proof_stream):
        """Stir::prove over the C ABI.  d_codeword: initial_domain.length XFE on the device.  Returns the first
        round's queried indices (what the STARK prover opens the trace at)."""
        from . import stark

        lib, ff = ctx.lib, self.folding_factor
        domain = self.initial_domain
        commitment = StirMerkleTree(ctx, d_codeword, domain.length, ff)
        proof_stream.enqueue("stir root", commitment.root())
        poly, n_coeffs = domain.interpolate(ctx, d_codeword, 3), domain.length
        first_round_indices = None
        self.rounds = []  # what a test wants to look at
        for in_domain, out_of_domain in self.round_queries:
            folding_randomness = proof_stream.sample_scalars(1)[0]
            folded, n_folded = fold_polynomial(ctx, poly, n_coeffs, ff, folding_randomness)
            next_domain = self.next_round_domain(domain)
            folded_evaluations = next_domain.evaluate(ctx, folded, n_folded, 3)
            folded_commitment = StirMerkleTree(ctx, folded_evaluations, next_domain.length, ff)
            proof_stream.enqueue("stir root", folded_commitment.root())

            ood_queries = proof_stream.sample_scalars(out_of_domain)
            ood_values = stark.evaluate_at_points(ctx, folded, n_folded, ood_queries)
            proof_stream.enqueue("stir ood values", ood_values)

            queried_indices = proof_stream.sample_indices(domain.length, in_domain)
            folded_domain = domain.pow(ff)
            folded_queried = list(dict.fromkeys(i % folded_domain.length for i in queried_indices))  # .unique()
            leafs, auth = commitment.inclusion_proof(folded_queried)
            proof_stream.enqueue("stir response leafs", leafs, fiat_shamir=False)
            proof_stream.enqueue("stir response auth", auth, fiat_shamir=False)

            # the witness polynomial of the next round
            queried_domain_values = np.zeros((len(folded_queried), 3), np.uint64)
            queried_domain_values[:, 0] = [folded_domain.value(i) for i in folded_queried]
            # folded_poly.evaluate at the queried points of the folded domain (stir.rs:946-950): one transform onto
            # that domain and a gather instead of ~200 Horner passes over 2^21 coefficients
            on_folded_domain = folded_domain.evaluate(ctx, folded, n_folded, 3)
            fq = np.array(folded_queried, np.uint64)
            in_domain_answers = np.empty((fq.size, 3), np.uint64)
            ctx._check(lib.tvm_gather_elements(ctx.handle, on_folded_domain.ptr, 3, fq.ctypes.data, fq.size,
                                               in_domain_answers.ctypes.data), "stir answers")
            del on_folded_domain
            quotient_answers = np.concatenate([in_domain_answers, ood_values])
            quotient_set = np.ascontiguousarray(np.concatenate([queried_domain_values, ood_queries]))
            k = quotient_set.shape[0]
            answer_poly = np.empty((k, 3), np.uint64)
            quotient_answers = np.ascontiguousarray(quotient_answers)
            if lib.tvm_xfe_interpolate(ctx.handle, quotient_set.ctypes.data, quotient_answers.ctypes.data, k, answer_poly.ctypes.data):
                raise ValueError("STIR quotient set has repeated points")   # (on the device: the host form leaves it idle ~1 ms)
            degree_correction_randomness = proof_stream.sample_scalars(1)[0]
            # any coset of >= n_folded points that avoids the quotient set: 7 generates F_p^*, so 7 * offset * <w>
            # is disjoint from offset * <w'> for every 2-power subgroup; the out-of-domain points are not in F_p
            work = ArithmeticDomain.of_length(n_folded).with_offset(field.mont_mul(folded_domain.offset, field.generator()))
            nxt = ctx.alloc(3 * n_folded)
            rc = np.ascontiguousarray(degree_correction_randomness, dtype=np.uint64)
            ctx._check(lib.tvm_stir_next_polynomial(ctx.handle, folded.ptr, n_folded, quotient_set.ctypes.data,
                                                    answer_poly.ctypes.data, k, rc.ctypes.data, work.c(), nxt.ptr),
                       "stir_next_polynomial")
            self.rounds.append(dict(folding_randomness=folding_randomness, domain=next_domain, root=folded_commitment.root(),
                                    ood_queries=ood_queries, ood_values=ood_values, queried_indices=queried_indices,
                                    folded_queried=folded_queried, quotient_set=quotient_set,
                                    quotient_answers=quotient_answers, degree_correction_randomness=rc))
            poly, n_coeffs = nxt, n_folded
            domain, commitment = next_domain, folded_commitment
            if first_round_indices is None:
                first_round_indices = queried_indices

        # the final round has no quotienting
        folding_randomness = proof_stream.sample_scalars(1)[0]
        final, n_final = fold_polynomial(ctx, poly, n_coeffs, ff, folding_randomness)
        self.final_folding_randomness = folding_randomness
        self.final_polynomial = final.download((n_final, 3))
        proof_stream.enqueue("stir final polynomial", self.final_polynomial)
        folded_domain = domain.pow(ff)
        queried_indices = proof_stream.sample_indices(domain.length, self.final_num_in_domain_queries)
        folded_queried = list(dict.fromkeys(i % folded_domain.length for i in queried_indices))
        leafs, auth = commitment.inclusion_proof(folded_queried)
        proof_stream.enqueue("stir response leafs", leafs, fiat_shamir=False)
        proof_stream.enqueue("stir response auth", auth, fiat_shamir=False)
        return first_round_indices if first_round_indices is not None else queried_indices


def fold_polynomial(ctx, d_poly, n_coeffs, folding_factor, randomness):
    """Stir::fold_polynomial (stir.rs:1132-1147) -> (device polynomial, its number of coefficients)"""
    n_out = -(-n_coeffs // folding_factor)
    out = ctx.alloc(3 * max(n_out, 1))
    r = np.ascontiguousarray(randomness, dtype=np.uint64)
    ctx._check(ctx.lib.tvm_fold_polynomial(ctx.handle, d_poly.ptr, n_coeffs, folding_factor, r.ctypes.data, out.ptr), "fold")
    return out, n_out


class StirMerkleTree:
    """StirMerkleTree (stir.rs:1380-1419): leaves are stacks of `stack_height` codeword entries taken at distance
    len / stack_height; the codeword and the node array stay on the device."""

    def __init__(self, ctx, d_codeword, length, stack_height):
        self.ctx, self.d_codeword, self.length, self.stack_height = ctx, d_codeword, length, stack_height
        self.n_leaves = length // stack_height
        self.d_nodes = ctx.alloc(10 * self.n_leaves)
        ctx._check(ctx.lib.tvm_stir_merkle_tree(ctx.handle, d_codeword.ptr, length, stack_height, self.d_nodes.ptr),
                   "stir_merkle_tree")

    def root(self):
        from . import stark

        return stark.merkle_root(self.ctx, self.d_nodes)

    def inclusion_proof(self, indices):
        """StirMerkleTree::inclusion_proof (stir.rs:1421-1440): the queried stacked leafs and the sibling nodes"""
        from . import stark

        idx = np.array([i + j * self.n_leaves for i in indices for j in range(self.stack_height)], np.uint64)
        leafs = np.empty((idx.size, 3), np.uint64)
        if idx.size:
            self.ctx._check(self.ctx.lib.tvm_gather_elements(self.ctx.handle, self.d_codeword.ptr, 3, idx.ctypes.data, idx.size,
                                                             leafs.ctypes.data), "stir leafs")
        return (leafs.reshape(len(indices), self.stack_height, 3),
                stark.auth_nodes(self.ctx, self.d_nodes, self.n_leaves, indices))
