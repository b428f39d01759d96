"""TEST INFRASTRUCTURE (oracle): `Prover::prove` with the reference's REAL transcript -- `Claim`, `ProofStream`,
`BFieldCodec`, Fiat-Shamir sampling, the prover's seeded randomness -- restated end to end.  It reproduces BOTH
proof-digest snapshots the reference holds (tests/test_proof_snapshot.py):
    current_proof_version_is_still_current   /root/reference/triton-vm/src/proof.rs:200-226
    supplying_prover_randomness_seed_fully_derandomizes_produced_proof   /root/reference/triton-vm/src/stark.rs:2434-2460
Everything in-tree is followed line by line (stark.rs:331-719 prove, low_degree_test/fri.rs:212-345 + 754-920 the FRI
prover and its parameters, proof_item.rs, proof_stream.rs, master_table.rs:392-434 + 612-662 + 1006-1030 randomness).
What lives only in `twenty-first = "2.0.0"` / `rand` (not vendored) is restated from its published behaviour; the
`Variant` knobs name those behaviours, and their defaults are the ones the snapshots confirm: derive(BFieldCodec)
encodes a struct's fields last-field-first with length prefixes on dynamically-sized fields, an enum as discriminant +
fields, Polynomials without trailing zeros; MerkleTree::authentication_structure is in descending node order;
`rng.random::<[u8; 32]>()` draws one u32 per byte; `Tip5::hash(&proof)` hashes the struct encoding of `Proof`.
Plain python + the C oracle; small traces only.  The product never imports this file."""
import itertools

import numpy as np

from . import degree_lowering as dlo
from . import oracle as orc
from . import ref_rng as rr
from .vm import tables as T, vm

P = orc.P
NUM_MAIN, NUM_AUX = 379, 91
ZETA = 3
PROOF_VERSION = 6                      # proof.rs:33 CURRENT_VERSION


def fri_num_collinearity_checks(security_level, log2_expansion):
    """fri.rs:832-836 with ReedSolomonCode::proximity_parameter for proven soundness (low_degree_test/mod.rs:93-170):
    the proximity margin is sqrt(rate), the slackness a twentieth of it"""
    import math

    margin = math.sqrt(1.0 / (1 << log2_expansion))
    return math.ceil(-security_level / math.log2(1.0 - (1.0 - margin - margin / 20.0)))


# ---- small field helpers on Montgomery words -------------------------------------------------------------------------------
def M(v):
    return int(orc.bfe(v))


def mont_list(values):
    return [int(x) for x in orc.to_mont(np.array([int(v) % P for v in values], dtype=object))] if len(values) else []


class Variant:
    """the uncertain [twenty-first] behaviours"""

    def __init__(self, struct_reversed=True, enum_prefix_dynamic=True, auth_descending=True, seed_per_u32=True,
                 proof_hash_struct_framing=True, normalize_polynomial=True, vec_dynamic_elem_prefix=True):
        self.struct_reversed = struct_reversed
        self.enum_prefix_dynamic = enum_prefix_dynamic
        self.auth_descending = auth_descending
        self.seed_per_u32 = seed_per_u32
        self.proof_hash_struct_framing = proof_hash_struct_framing
        self.normalize_polynomial = normalize_polynomial
        self.vec_dynamic_elem_prefix = vec_dynamic_elem_prefix

    def __repr__(self):
        return "Variant(" + ", ".join(f"{k}={v}" for k, v in self.__dict__.items()) + ")"


# ---- BFieldCodec (encodings are lists of CANONICAL integers) ------------------------------------------------------------
class Enc:
    """an encoded value: words + whether its type has a static length"""

    def __init__(self, words, static):
        self.words, self.static = list(words), static


def enc_bfe_words(words):
    return Enc([int(w) for w in words], True)


def enc_vec_static(elements_words, n_elements):
    """Vec<T>, T of static length: [number of elements, elements...]"""
    return Enc([n_elements] + list(elements_words), False)


def enc_vec_dynamic(encs, variant):
    out = [len(encs)]
    for e in encs:
        if variant.vec_dynamic_elem_prefix:
            out.append(len(e.words))
        out += e.words
    return Enc(out, False)


def enc_struct(fields, variant):
    """derive(BFieldCodec) on a struct: the fields, each dynamic one prefixed with its length"""
    fields = list(reversed(fields)) if variant.struct_reversed else list(fields)
    out = []
    for f in fields:
        if not f.static:
            out.append(len(f.words))
        out += f.words
    return Enc(out, all(f.static for f in fields))


VARIANTS_OF_PROOF_ITEM = ["MerkleRoot", "Log2PaddedHeight", "OutOfDomainMainRow", "OutOfDomainAuxRow", "OutOfDomainQuotientSegments",
                          "Polynomial", "StirOutOfDomainValues", "AuthenticationStructure", "MasterMainTableRows", "MasterAuxTableRows",
                          "QuotientSegmentsElements", "FriCodeword", "FriResponse", "StirResponse"]
IN_FIAT_SHAMIR = {"MerkleRoot", "Log2PaddedHeight", "OutOfDomainMainRow", "OutOfDomainAuxRow", "OutOfDomainQuotientSegments", "Polynomial",
                  "StirOutOfDomainValues"}


def enc_proof_item(name, payload, variant):
    """derive(BFieldCodec) on the enum ProofItem (proof_item.rs:19-160): discriminant, then the payload"""
    out = [VARIANTS_OF_PROOF_ITEM.index(name)]
    if not payload.static and variant.enum_prefix_dynamic:
        out.append(len(payload.words))
    return Enc(out + payload.words, False)


# ---- the sponge --------------------------------------------------------------------------------------------------------------
class Sponge:
    """Tip5 in variable-length mode (Tip5::init): overwrite-mode absorb, squeeze = rate then permutation"""

    def __init__(self):
        self.state = np.zeros(16, np.uint64)

    def pad_and_absorb_all(self, words_canonical):
        words = mont_list(words_canonical)
        padded = words + [M(1)] + [0] * ((-len(words) - 1) % 10)
        for k in range(0, len(padded), 10):
            self.state[:10] = np.array(padded[k:k + 10], np.uint64)
            self.state = orc.tip5_permutation(self.state)

    def squeeze(self):
        out = [int(w) for w in self.state[:10]]
        self.state = orc.tip5_permutation(self.state)
        return out

    def sample_scalars(self, n):
        words = []
        for _ in range((3 * n + 9) // 10):
            words += self.squeeze()
        return [np.array(words[3 * i:3 * i + 3], np.uint64) for i in range(n)]

    def sample_indices(self, upper_bound, n):
        out, pending = [], []
        while len(out) < n:
            if not pending:
                pending = self.squeeze()
            w = pending.pop(0)
            v = int(orc.value(w))
            if v != P - 1:
                out.append(v % upper_bound)
        return out


class ProofStream:
    def __init__(self, variant):
        self.items, self.sponge, self.variant = [], Sponge(), variant

    def alter_fiat_shamir_state_with(self, enc):
        self.sponge.pad_and_absorb_all(enc.words)

    def enqueue(self, name, payload):
        item = enc_proof_item(name, payload, self.variant)
        if name in IN_FIAT_SHAMIR:
            self.alter_fiat_shamir_state_with(item)
        self.items.append(item)

    def sample_scalars(self, n):
        return self.sponge.sample_scalars(n)

    def sample_indices(self, upper_bound, n):
        return self.sponge.sample_indices(upper_bound, n)

    def proof_words(self):
        """Proof(proof_stream.encode()): ProofStream is a struct whose only encoded field is `items`"""
        return enc_struct([enc_vec_dynamic(self.items, self.variant)], self.variant).words


# ---- Merkle trees ------------------------------------------------------------------------------------------------------------
def authentication_structure(nodes, n_leaves, indices, variant):
    """MerkleTree::authentication_structure [twenty-first]: the nodes needed besides the revealed leaves, no duplicates"""
    needed, computable = set(), set()
    for i in indices:
        node = n_leaves + i
        while node > 1:
            needed.add(node ^ 1)
            computable.add(node)
            node >>= 1
    order = sorted(needed - computable, reverse=variant.auth_descending)
    return [nodes[k] for k in order]


def values_of(words):
    return [int(v) for v in orc.from_mont(np.asarray(words, np.uint64).reshape(-1))]


def enc_xfes(xfes):
    return values_of(np.asarray(xfes, np.uint64))


def enc_digests_vec(digests):
    return enc_vec_static(values_of(np.asarray(digests, np.uint64)) if len(digests) else [], len(digests))


# ---- randomness ----------------------------------------------------------------------------------------------------------------
def offset_rng_seed(seed, offset):
    """master_table.rs:631-662"""
    seed = list(seed)
    add = list(int(offset).to_bytes(8, "little")) + [0] * 24
    carry = 0
    for k in range(32):
        s = seed[k] + add[k] + carry
        seed[k], carry = s & 0xFF, s >> 8
    return bytes(seed)


def random_elements(seed, n, fk):
    rng = rr.StdRng.from_seed(seed)
    vals = [[rng.range_canon() for _ in range(fk)] for _ in range(n)]
    a = orc.to_mont(np.array(vals, dtype=object))
    return a.reshape(n) if fk == 1 else a


def xfe_pow(x, e):
    return orc.xfe_pow(x, e)


def xfe_scale(x, b_mont):
    return np.array([orc.lib().orc_bfe_mul(int(c), int(b_mont)) for c in x], np.uint64)


# ---- the prover ----------------------------------------------------------------------------------------------------------------
def stir_prove(ps, codeword, ldt, stir, variant):
    """Stir::prove (low_degree_test/stir.rs:885-993), statement by statement, on polynomials in COEFFICIENT form with the arithmetic of
    oracle/stir_oracle.py -- Lagrange interpolation, an explicit zerofier, long division, schoolbook multiplication: none of what
    csrc/stir.hip does (which works on evaluations over a coset and never divides polynomials).  `stir`: the instance's numbers
    (folding_factor, round_queries [(in_domain, out_of_domain)], final_num_in_domain_queries) -- the parameter derivation is host
    arithmetic pinned by the reference's own constants (tests/test_stir_parameters.py), not part of what is compared here.
    Returns the first round's queried indices (the rows the STARK prover opens)."""
    from . import stir_oracle as so

    ff = stir["folding_factor"]
    domain = ldt
    stacks, nodes = so.stack_tree(codeword, ff)
    ps.enqueue("MerkleRoot", enc_bfe_words(values_of(nodes[1])))
    poly = orc.coset_interpolate(codeword, domain, 3).reshape(-1, 3)
    first_round_queried_indices = None

    def next_round_domain(d):   # stir.rs:1149-1155: the squares of the domain's points, shifted by the domain's offset
        nxt = orc.domain_pow(d, 2)
        nxt.offset = int(orc.lib().orc_bfe_mul(nxt.offset, d.offset))
        return nxt

    def unique(seq):
        return list(dict.fromkeys(seq))

    def respond(stacks, nodes, n_leaves, folded_indices):
        # StirMerkleTree::inclusion_proof (stir.rs:1421-1440) -> StirResponse {queried_leafs: Vec<Vec<XFE>>, auth_structure}
        leafs = [enc_vec_static(enc_xfes(list(stacks[i])), len(stacks[i])) for i in folded_indices]
        auth = authentication_structure(nodes, n_leaves, folded_indices, variant)
        ps.enqueue("StirResponse", enc_struct([enc_vec_dynamic(leafs, variant), enc_digests_vec(auth)], variant))

    for in_domain, out_of_domain in stir["round_queries"]:
        folding_randomness = ps.sample_scalars(1)[0]
        folded = so.fold_polynomial(poly, ff, folding_randomness)
        nxt_domain = next_round_domain(domain)
        folded_evaluations = orc.coset_evaluate(folded, nxt_domain, 3).reshape(-1, 3)
        nxt_stacks, nxt_nodes = so.stack_tree(folded_evaluations, ff)
        ps.enqueue("MerkleRoot", enc_bfe_words(values_of(nxt_nodes[1])))

        ood_queries = ps.sample_scalars(out_of_domain)
        ood_values = [orc.poly_eval_xfe(folded, x) for x in ood_queries]
        ps.enqueue("StirOutOfDomainValues", enc_vec_static(enc_xfes(ood_values), len(ood_values)))

        queried_indices = ps.sample_indices(domain.length, in_domain)
        folded_domain = orc.domain_pow(domain, ff)
        folded_queried = unique(i % folded_domain.length for i in queried_indices)
        respond(stacks, nodes, domain.length // ff, folded_queried)

        # the witness polynomial of the next round
        values = orc.domain_values(folded_domain)
        queried_domain_values = [np.array([values[i], 0, 0], np.uint64) for i in folded_queried]
        answers = [orc.poly_eval_xfe(folded, x) for x in queried_domain_values] + ood_values
        quotient_set = queried_domain_values + list(ood_queries)
        degree_correction_randomness = ps.sample_scalars(1)[0]
        poly = np.array(so.next_polynomial(folded, quotient_set, answers, degree_correction_randomness,
                                           interpolate=so.lagrange_interpolate_from_zerofier), np.uint64).reshape(-1, 3)
        domain, stacks, nodes = nxt_domain, nxt_stacks, nxt_nodes
        if first_round_queried_indices is None:
            first_round_queried_indices = queried_indices

    # the final round: no quotienting
    folding_randomness = ps.sample_scalars(1)[0]
    final = so.fold_polynomial(poly, ff, folding_randomness)
    coeffs = [values_of(c) for c in final]
    if variant.normalize_polynomial:
        while coeffs and coeffs[-1] == [0, 0, 0]:
            coeffs.pop()
    ps.enqueue("Polynomial", enc_struct([enc_vec_static(list(itertools.chain.from_iterable(coeffs)), len(coeffs))], variant))
    folded_domain = orc.domain_pow(domain, ff)
    queried_indices = ps.sample_indices(domain.length, stir["final_num_in_domain_queries"])
    respond(stacks, nodes, domain.length // ff, unique(i % folded_domain.length for i in queried_indices))
    return first_round_queried_indices if first_round_queried_indices is not None else queried_indices


def prove(program, public_input, secret_input=(), secret_digests=(), ram=None, seed_u64=None, variant=None, security_level=160, stir=None,
          spill_dir=None):
    """`stir`: None = LdtChoice::Fri (what the reference's snapshots use: FRI-sized programs), or the numbers of the STIR instance
    (dict: initial_domain_length, num_trace_randomizers, folding_factor, round_queries, final_num_in_domain_queries) = LdtChoice::Stir.
    `spill_dir`: a directory for the two extended tables as file-backed arrays (2^19 rows and more: 22 / 44 GB of tables next to the
    Python-integer trace tables do not fit a 64 GB host); same results."""

    def table_array(name, shape):
        if spill_dir is None:
            return None
        import os

        return np.lib.format.open_memmap(os.path.join(spill_dir, name + ".npy"), mode="w+", dtype=np.uint64, shape=shape)

    variant = variant or Variant()
    aet, output = vm.trace_execution(program, public_input, secret_input, secret_digests, ram)
    program_digest = vm.hash_varlen(program.to_bwords())
    public_input = [v % P for v in public_input]
    # the seed: rng.random::<[u8; 32]>() of StdRng::seed_from_u64 (proof.rs:212-214)
    rng = rr.StdRng.seed_from_u64(seed_u64)
    seed = bytes((rng.next_u32() & 0xFF) for _ in range(32)) if variant.seed_per_u32 else rng.fill_bytes(32)

    ps = ProofStream(variant)
    claim = enc_struct([enc_bfe_words(program_digest), enc_bfe_words([PROOF_VERSION]), enc_vec_static(public_input, len(public_input)),
                        enc_vec_static(output, len(output))], variant)
    ps.alter_fiat_shamir_state_with(claim)

    # parameters: Stark::default() (stark.rs:1885-2089), FRI below 2^16 rows (fri.rs:797-920)
    import math

    num_checks = fri_num_collinearity_checks(security_level, 2)
    h = num_checks + 4 * 3 * 2 + 1 if stir is None else stir["num_trace_randomizers"]
    padded_height = aet.padded_height()
    rtl = 1 << (max(padded_height + h, 2 * h + 1, (h + 1) * 5) - 1).bit_length()
    n = rtl // 2
    log2_hdb = padded_height.bit_length() - 1
    while True:
        log2_hdb += 1
        ldt_len = 1 << (log2_hdb + 2)
        if ldt_len >= rtl * 4:
            break
    if stir is not None:
        ldt_len = stir["initial_domain_length"]   # Stark::stir (stark.rs:1972-2032) sizes the domain itself
    quot_len = 4 * rtl          # max_degree = 4 * rtl - 1 for every padded height (stark.rs:1905-1916)
    assert quot_len == ldt_len, "only the shape quotient domain == LDT domain is restated here"
    g = orc.lib().orc_bfe_generator()
    trace_dom = orc.domain_of_length(n)
    ldt = orc.domain_of_length(ldt_len, offset=g)
    quot = orc.domain_of_length(quot_len, offset=g)
    fri_rounds = max(0, (ldt_len // 4).bit_length() - 1 - (num_checks.bit_length() - 1 + 1))
    ps.enqueue("Log2PaddedHeight", enc_bfe_words([padded_height.bit_length() - 1]))

    # main table: fill, pad (to the trace domain's length), degree lowering; randomized LDE; commitment
    mt = T.MasterMainTable(aet, padded_height=n).pad()
    main = np.zeros((NUM_MAIN, n), np.uint64)
    main[:T.NUM_MAIN] = orc.to_mont(np.array(mt.columns(), dtype=object))
    main, _ = dlo.fill(main)
    main_rnd = np.stack([random_elements(offset_rng_seed(seed, c), h, 1) for c in range(NUM_MAIN)])
    main_lde = orc.lde_table(main, main_rnd, ldt, 1, out=table_array("main_lde", (ldt_len, NUM_MAIN)))
    main_nodes = orc.merkle_tree(orc.hash_rows(main_lde))
    ps.enqueue("MerkleRoot", enc_bfe_words(values_of(main_nodes[1])))
    sampled = ps.sample_scalars(59)
    challenges = T.derive_challenges([values_of(c) for c in sampled], program_digest, public_input, output)
    ch = orc.to_mont(np.array(challenges, dtype=object))

    # aux table: extend, batch randomizer column, degree lowering (master_table.rs:1006-1075)
    aux_seed = offset_rng_seed(seed, NUM_MAIN)
    aux = np.zeros((NUM_AUX, n, 3), np.uint64)
    aux[:T.NUM_AUX] = orc.to_mont(np.array(T.extend(mt.tables, challenges), dtype=object))
    aux[NUM_AUX - 1] = random_elements(offset_rng_seed(aux_seed, NUM_AUX), n, 3)
    _, aux = dlo.fill(main, aux, ch)
    aux_rnd = np.stack([random_elements(offset_rng_seed(aux_seed, c), h, 3) for c in range(NUM_AUX)])
    aux_lde = orc.lde_table(aux, aux_rnd, ldt, 3, out=table_array("aux_lde", (ldt_len, NUM_AUX, 3)))
    aux_nodes = orc.merkle_tree(orc.hash_rows(aux_lde.reshape(ldt_len, -1)))
    ps.enqueue("MerkleRoot", enc_bfe_words(values_of(aux_nodes[1])))
    w0 = ps.sample_scalars(1)[0]
    quotient_weights = np.array([xfe_pow(w0, i) for i in range(604)], np.uint64)

    # quotient: codeword, segments, randomization, commitment (stark.rs:405-446, 1224-1356)
    q = orc.quotients_combined(main_lde, aux_lde, trace_dom, quot, ch, quotient_weights)
    seg = orc.interpolate_quotient_segments(q, quot)
    quot_rnd = random_elements(offset_rng_seed(seed, NUM_MAIN + NUM_AUX + 1), (h + 1) * 5, 3)
    polys, seg_cws = orc.randomize_quotient_segments(seg, quot_rnd, ldt)
    quot_nodes = orc.merkle_tree(orc.hash_rows(seg_cws.reshape(ldt_len, 15)))
    ps.enqueue("MerkleRoot", enc_bfe_words(values_of(quot_nodes[1])))

    # out-of-domain rows (stark.rs:450-495)
    alpha = ps.sample_scalars(1)[0]
    alpha_next = xfe_scale(alpha, trace_dom.generator)
    for point in (alpha, alpha_next):
        ps.enqueue("OutOfDomainMainRow", enc_bfe_words(enc_xfes(orc.out_of_domain_row(main, main_rnd, point, 1))))
        ps.enqueue("OutOfDomainAuxRow", enc_bfe_words(enc_xfes(orc.out_of_domain_row(aux, aux_rnd, point, 3))))
    a4 = xfe_pow(alpha, 4)
    za4 = xfe_pow(xfe_scale(alpha, M(ZETA)), 4)
    seg_p = [orc.poly_eval_xfe(polys[k], a4) for k in range(4)]
    seg_r = [orc.poly_eval_xfe(polys[k], za4) for k in range(1, 5)]
    ps.enqueue("OutOfDomainQuotientSegments", enc_bfe_words(enc_xfes(seg_p)))
    ps.enqueue("OutOfDomainQuotientSegments", enc_bfe_words(enc_xfes(seg_r)))

    # combination codeword (stark.rs:497-640)
    iw = ps.sample_scalars(3)
    wm = np.array([xfe_pow(iw[0], i) for i in range(NUM_MAIN + NUM_AUX)], np.uint64)
    wq = np.array([xfe_pow(iw[1], i) for i in range(5)], np.uint64)
    wd = np.array([xfe_pow(iw[2], i) for i in range(4)], np.uint64)
    comb = orc.weighted_sum_of_columns(main, main_rnd, wm[:NUM_MAIN], 1)
    comb_aux = orc.weighted_sum_of_columns(aux, aux_rnd, wm[NUM_MAIN:], 3)
    comb = np.array([orc.xfe_add(a, b) for a, b in zip(comb, comb_aux)], np.uint64)
    ma_cw = orc.coset_evaluate(comb, ldt, 3).reshape(-1, 3)

    def xsum(terms):
        acc = np.zeros(3, np.uint64)
        for t in terms:
            acc = orc.xfe_add(acc, t)
        return acc

    plen = polys.shape[1]
    p_poly = np.array([xsum([orc.xfe_mul(wq[k], polys[k, j]) for k in range(4)]) for j in range(plen)])
    r_poly = np.array([xsum([orc.xfe_mul(wq[k], polys[k, j]) for k in range(1, 5)]) for j in range(plen)])
    p_cw = orc.coset_evaluate(p_poly, ldt, 3).reshape(-1, 3)
    r_cw = orc.coset_evaluate(r_poly, ldt, 3).reshape(-1, 3)
    parts = [orc.deep_codeword(ma_cw, ldt, alpha, orc.poly_eval_xfe(comb, alpha)),
             orc.deep_codeword(ma_cw, ldt, alpha_next, orc.poly_eval_xfe(comb, alpha_next)),
             orc.deep_codeword(p_cw, ldt, a4, orc.poly_eval_xfe(p_poly, a4)),
             orc.deep_codeword(r_cw, ldt, za4, orc.poly_eval_xfe(r_poly, za4))]
    codeword = np.array([xsum([orc.xfe_mul(parts[k][i], wd[k]) for k in range(4)]) for i in range(ldt_len)], np.uint64)

    if stir is not None:
        a_indices = stir_prove(ps, codeword, ldt, stir, variant)
    else:
        # FRI (fri.rs:130-345, 757-775)
        rounds = []
        dom, cw = ldt, codeword
        for r in range(fri_rounds + 1):
            if r:
                challenge = ps.sample_scalars(1)[0]
                cw = orc.fri_split_and_fold(cw, dom, challenge)
                dom = orc.domain_pow(dom, 2)
            nodes = orc.merkle_tree(orc.xfe_to_digest(cw))
            ps.enqueue("MerkleRoot", enc_bfe_words(values_of(nodes[1])))
            rounds.append((dom, cw, nodes))
        last_cw = rounds[-1][1]
        ps.enqueue("FriCodeword", enc_vec_static(enc_xfes(last_cw), len(last_cw)))
        last_poly = orc.coset_interpolate(last_cw, orc.domain_of_length(len(last_cw)), 3).reshape(-1, 3)
        coeffs = [values_of(c) for c in last_poly]
        if variant.normalize_polynomial:
            while coeffs and coeffs[-1] == [0, 0, 0]:
                coeffs.pop()
        poly_enc = enc_struct([enc_vec_static(list(itertools.chain.from_iterable(coeffs)), len(coeffs))], variant)
        ps.enqueue("Polynomial", poly_enc)
        a_indices = ps.sample_indices(ldt_len, num_checks)

        def respond(r, indices):
            d, c, nodes = rounds[r]
            leaves = [c[i] for i in indices]
            auth = authentication_structure(nodes, d.length, indices, variant)
            fields = [enc_vec_static(enc_xfes(leaves), len(leaves)), enc_digests_vec(auth)]   # queried_leaves, auth_structure
            ps.enqueue("FriResponse", enc_struct(fields, variant))

        respond(0, a_indices)
        for r in range(len(rounds) - 1):
            n_r = rounds[r][0].length
            respond(r, [(i + n_r // 2) % n_r for i in a_indices])
        ps.sample_scalars(1)

    # open the trace leafs (stark.rs:665-716)
    for name, lde, nodes, width in (("MasterMainTableRows", main_lde, main_nodes, NUM_MAIN),
                                    ("MasterAuxTableRows", aux_lde.reshape(ldt_len, -1), aux_nodes, NUM_AUX * 3),
                                    ("QuotientSegmentsElements", seg_cws.reshape(ldt_len, 15), quot_nodes, 15)):
        rows = [values_of(lde[i]) for i in a_indices]
        ps.enqueue(name, enc_vec_static(list(itertools.chain.from_iterable(rows)), len(rows)))
        ps.enqueue("AuthenticationStructure", enc_digests_vec(authentication_structure(nodes, ldt_len, a_indices, variant)))

    proof = ps.proof_words()
    hashed = enc_struct([enc_vec_static(proof, len(proof))], variant).words if variant.proof_hash_struct_framing else [len(proof)] + proof
    digest = values_of(orc.hash_varlen(orc.to_mont(np.array(hashed, dtype=object))))
    return {"proof": proof, "digest": digest, "claim": claim.words, "indices": a_indices, "params": dict(n=n, h=h, ldt=ldt_len, checks=num_checks,
                                                                                                     fri_rounds=fri_rounds)}
