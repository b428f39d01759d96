"""Pin the CPU oracle against every known-answer value the reference tree holds for the hot path
 and against an independent big-int Python restatement of the spec text.

Pins:
  * Montgomery encoding 42 <-> 180388626390   (triton-constraint-builder/src/codegen.rs:926-944)
  * Tip5 fixed-length chain                    (tips/tip-0005/tip-0005.md:394-405)
  * Tip5 variable-length vectors, 0..9 words   (tips/tip-0005/tip-0005.md:410-419)
    -- the 10..19-word vectors (:420-430) describe an *additive* sponge and are stale; the code
       overwrites the rate (specification/src/hash-table.md:24-26, master_table.rs:667-716).
  * NTT convention: evaluate == Horner at domain.values()  (arithmetic_domain.rs:361-393,457-473)
  * LDE sub-sampling identity                               (arithmetic_domain.rs:395-415)
  * zerofier identities                                     (arithmetic_domain.rs:475-489)
"""
import numpy as np
import pytest

P = 2**64 - 2**32 + 1
R = 2**64 % P
RINV = pow(R, -1, P)


# ---------------------------------------------------------------- independent big-int mirror
def py_mont(v):
    return v * R % P


def py_tip5_tables():
    from tools.gen_tip5_constants import LOOKUP, MDS_FIRST_COLUMN, round_constants_montgomery_raw

    return [c * RINV % P for c in round_constants_montgomery_raw()], MDS_FIRST_COLUMN, LOOKUP


def py_perm(st):
    rc, mds, lut = py_tip5_tables()

    def sbox(x):
        raw = x * R % P
        out = bytes(lut[b] for b in raw.to_bytes(8, "little"))
        return int.from_bytes(out, "little") * RINV % P

    for r in range(5):
        st = [sbox(st[i]) if i < 4 else pow(st[i], 7, P) for i in range(16)]
        st = [sum(mds[(i - j) % 16] * st[j] for j in range(16)) % P for i in range(16)]
        st = [(st[i] + rc[r * 16 + i]) % P for i in range(16)]
    return st


def py_hash_varlen(inp):
    inp = list(inp) + [1]
    inp += [0] * ((-len(inp)) % 10)
    st = [0] * 16
    for i in range(0, len(inp), 10):
        st[:10] = inp[i:i + 10]
        st = py_perm(st)
    return st[:5]


# ---------------------------------------------------------------- tests
def test_montgomery_kat(orc):
    assert orc.bfe(42) == 180388626390
    assert orc.value(180388626390) == 42
    assert orc.bfe(1) == 2**32 - 1


def test_field_ops_match_bigint(orc):
    rng = np.random.default_rng(1)
    L = orc.lib()
    edge = [0, 1, P - 1, 2**32 - 1, 2**32, 2**63, P - 2**32]
    vals = edge + [int(x) for x in rng.integers(0, P, 40, dtype=np.uint64)]
    for a in vals:
        for b in vals[:12]:
            ra, rb = py_mont(a), py_mont(b)
            assert L.orc_bfe_mul(ra, rb) == py_mont(a * b % P)
            assert L.orc_bfe_add(ra, rb) == py_mont((a + b) % P)
            assert L.orc_bfe_sub(ra, rb) == py_mont((a - b) % P)
        if a:
            assert L.orc_bfe_inv(py_mont(a)) == py_mont(pow(a, -1, P))


def test_xfe_inverse_and_mul(orc):
    rng = np.random.default_rng(2)
    one = np.array([orc.bfe(1), 0, 0], np.uint64)
    for _ in range(20):
        a = orc.random_elements(rng, 3)
        assert (orc.xfe_mul(a, orc.xfe_inv(a)) == one).all()
    # X * X^2 = X^3 = X - 1
    x = np.array([0, orc.bfe(1), 0], np.uint64)
    x2 = orc.xfe_mul(x, x)
    assert list(orc.xfe_mul(x, x2)) == [orc.bfe(P - 1), orc.bfe(1), 0]
    # Fermat in the extension: a^(p^3 - 1) = 1 is too slow; check the norm lands in F_p instead
    a = orc.random_elements(rng, 3)
    ap = a.copy()
    for _ in range(1):
        ap = orc.xfe_pow(ap, P)          # Frobenius
    app = orc.xfe_pow(ap, P)
    norm = orc.xfe_mul(orc.xfe_mul(a, ap), app)
    assert norm[1] == 0 and norm[2] == 0


def test_roots_of_unity(orc):
    L = orc.lib()
    # consistent with SURVEY 8c: 2^32-th root = 7^((p-1)/2^32)
    assert orc.value(L.orc_bfe_primitive_root(2**32)) == 1753635133440165772
    assert orc.value(L.orc_bfe_primitive_root(64)) == 2**39
    assert orc.value(L.orc_bfe_primitive_root(4)) == 2**48
    assert orc.value(L.orc_bfe_primitive_root(2)) == P - 1
    for lg in (1, 5, 10, 20):
        w = L.orc_bfe_primitive_root(1 << lg)
        assert L.orc_bfe_pow(w, 1 << lg) == orc.bfe(1)
        assert L.orc_bfe_pow(w, 1 << (lg - 1)) == orc.bfe(P - 1)


def test_tip5_fixed_length_chain(orc):
    """tips/tip-0005/tip-0005.md:394-405"""
    chain = [
        ([0] * 10, [941080798860502477, 5295886365985465639, 14728839126885177993, 10358449902914633406, 14220746792122877272]),
        ([941080798860502477, 5295886365985465639, 14728839126885177993, 10358449902914633406, 14220746792122877272, 0, 0, 0, 0, 0],
         [15888421881075650037, 8699648354187865464, 6719068786850902915, 16188941274693647820, 4768361305800190493]),
        ([941080798860502477, 15888421881075650037, 8699648354187865464, 6719068786850902915, 16188941274693647820, 4768361305800190493, 0, 0, 0, 0],
         [11494362724359741120, 2984169814429715553, 11021746812971026026, 5102281498552384717, 5023112854146751042]),
        ([941080798860502477, 15888421881075650037, 11494362724359741120, 2984169814429715553, 11021746812971026026, 5102281498552384717, 5023112854146751042, 0, 0, 0],
         [627201255727529993, 2530132417472465719, 15134374672529870482, 10586143339158028166, 13810271029904013559]),
        ([941080798860502477, 15888421881075650037, 11494362724359741120, 627201255727529993, 2530132417472465719, 15134374672529870482, 10586143339158028166, 13810271029904013559, 0, 0],
         [4790238723037855394, 13717377209729127271, 8994982932799814404, 18004412270774820131, 5877166878145340765]),
        ([941080798860502477, 15888421881075650037, 11494362724359741120, 627201255727529993, 4790238723037855394, 13717377209729127271, 8994982932799814404, 18004412270774820131, 5877166878145340765, 0],
         [16959020643814878453, 12118009629857908438, 10239930869937551135, 6889489196156760098, 5774309862903741805]),
        ([941080798860502477, 15888421881075650037, 11494362724359741120, 627201255727529993, 4790238723037855394, 16959020643814878453, 12118009629857908438, 10239930869937551135, 6889489196156760098, 5774309862903741805],
         [10869784347448351760, 1853783032222938415, 6856460589287344822, 17178399545409290325, 7650660984651717733]),
    ]
    for inp, want in chain:
        got = orc.from_mont(orc.hash_10(orc.to_mont(inp)))
        assert [int(x) for x in got] == want


VARLEN = {
    0: [2335476311349343808, 1307299401243390569, 3414029282375928929, 2141465175172981451, 5966553798353564426],
    1: [4843866011885844809, 16618866032559590857, 18247689143239181392, 7637465675240023996, 9104890367162237026],
    2: [14221897462292645957, 3690523333672640544, 7547831217417524560, 11517644941222042877, 16820478393376780897],
    3: [3557614275028747325, 18213566888269431883, 14211012637913216818, 18426990445135603349, 8015183961235958327],
    4: [13668806558765160443, 7736989284450687030, 15316066412582144917, 14566815392725049262, 1631258856522889875],
    5: [1380324360087351655, 2493688017679385677, 18197583438743680153, 2303632749506762680, 2500436438073253576],
    6: [1612925275097886605, 8293210493469698946, 5378029315601990928, 9997723552534409936, 18350405537085446855],
    7: [2368572306594843451, 13479396176400056076, 5509084167070310636, 9541200077614575285, 14698893519125746147],
    8: [5764047891359019962, 4580068493600531946, 6759906304791724061, 17885774121391644741, 5272177385407180638],
    9: [5188069162914592397, 852189275605886954, 1770154650497175879, 10044069521465249269, 15310276722084590255],
}


def test_tip5_varlen_vectors(orc):
    """tips/tip-0005/tip-0005.md:410-419 (single-block inputs; see module docstring for >= 10)."""
    for n, want in VARLEN.items():
        got = orc.from_mont(orc.hash_varlen(orc.to_mont(list(range(n))) if n else np.zeros(0, np.uint64)))
        assert [int(x) for x in got] == want, n


def test_tip5_multiblock_overwrite_mode_matches_bigint_mirror(orc):
    for n in (10, 11, 19, 20, 29, 45, 379):
        inp = [(7 * i + 3) % P for i in range(n)]
        got = [int(x) for x in orc.from_mont(orc.hash_varlen(orc.to_mont(inp)))]
        assert got == py_hash_varlen(inp), n


def test_permutation_matches_bigint_mirror(orc):
    rng = np.random.default_rng(5)
    for _ in range(3):
        st = [int(x) for x in rng.integers(0, P, 16, dtype=np.uint64)]
        got = [int(x) for x in orc.from_mont(orc.tip5_permutation(orc.to_mont(st)))]
        assert got == py_perm(st)


def test_evaluate_is_horner_on_domain_values(orc):
    """arithmetic_domain.rs:361-393: evaluate(poly)[i] == poly(domain.value(i)), natural order."""
    rng = np.random.default_rng(3)
    for lg, ncoef in ((3, 8), (5, 20), (4, 40)):       # last case exercises chunk folding :153-167
        d = orc.domain_of_length(1 << lg, offset=orc.lib().orc_bfe_generator())
        co = orc.random_elements(rng, ncoef)
        got = orc.from_mont(orc.coset_evaluate(co, d))
        xs = [int(v) for v in orc.from_mont(orc.domain_values(d))]
        cs = [int(v) for v in orc.from_mont(co)]
        for i, x in enumerate(xs):
            assert int(got[i]) == sum(c * pow(x, j, P) for j, c in enumerate(cs)) % P
    # interpolate inverts evaluate
    d = orc.domain_of_length(32, offset=orc.bfe(11))
    co = orc.random_elements(rng, 32)
    assert (orc.coset_interpolate(orc.coset_evaluate(co, d), d) == co).all()
    # and the XFE flavour is three interleaved base-field transforms
    cx = orc.random_elements(rng, (32, 3))
    ev = orc.coset_evaluate(cx, d, fk=3).reshape(32, 3)
    for k in range(3):
        assert (ev[:, k] == orc.coset_evaluate(np.ascontiguousarray(cx[:, k]), d)).all()


def test_low_degree_extension_subsampling(orc):
    """arithmetic_domain.rs:395-415: LDE onto a 4x longer domain with equal offset, every 4th
    value equals the original codeword."""
    rng = np.random.default_rng(4)
    short = orc.domain_of_length(16, offset=orc.bfe(5))
    long_ = orc.domain_of_length(64, offset=orc.bfe(5))
    cw = orc.random_elements(rng, 16)
    ext = orc.coset_evaluate(orc.coset_interpolate(cw, short), long_)
    assert (ext[::4] == cw).all()


def test_lde_table_restricts_to_trace_on_trace_domain(orc):
    """The randomized interpolant agrees with the trace on the trace domain
    (master_table.rs:392-403, specification/src/zero-knowledge.md:58-140) and has the
    randomizer's top coefficients at X^(N..N+h)."""
    rng = np.random.default_rng(6)
    n, h, ncols = 16, 5, 3
    trace = orc.random_elements(rng, (ncols, n))
    rnd = orc.random_elements(rng, (ncols, h))
    poly = orc.randomized_column_interpolant(trace[1], rnd[1])
    assert (poly[n:n + h] == rnd[1]).all() and (poly[n + h:] == 0).all()
    td = orc.domain_of_length(n)
    assert (orc.coset_evaluate(poly[:n + h], td) == trace[1]).all()
    ev = orc.domain_of_length(64, offset=orc.lib().orc_bfe_generator())
    table = orc.lde_table(trace, rnd, ev)
    assert table.shape == (64, ncols)
    assert (table[:, 1] == orc.coset_evaluate(poly[:n + h], ev)).all()


def test_zerofier_inverses(orc):
    """master_table.rs:1575-1625"""
    L = orc.lib()
    td = orc.domain_of_length(8)
    qd = orc.domain_of_length(32, offset=L.orc_bfe_generator())
    init, cons, tran, term = orc.zerofier_inverses(td, qd)
    xs = [int(v) for v in orc.from_mont(orc.domain_values(qd))]
    wi = pow(orc.value(td.generator), -1, P)
    for i, x in enumerate(xs):
        assert orc.value(init[i]) == pow(x - 1, -1, P)
        assert orc.value(cons[i]) == pow(pow(x, 8, P) - 1, -1, P)
        assert orc.value(tran[i]) == (x - wi) * pow(pow(x, 8, P) - 1, -1, P) % P
        assert orc.value(term[i]) == pow((x - wi) % P, -1, P)


def test_merkle_and_fold_shapes(orc):
    rng = np.random.default_rng(7)
    leaves = orc.random_elements(rng, (8, 5))
    nodes = orc.merkle_tree(leaves)
    assert (nodes[8:] == leaves).all()
    assert (nodes[1] == orc.hash_pair(nodes[2], nodes[3])).all()
    assert (nodes[5] == orc.hash_pair(leaves[2], leaves[3])).all()
    # FRI fold of a low-degree codeword stays low degree: fold(f)(x^2) = f_e(x^2) + c f_o(x^2)
    d = orc.domain_of_length(32, offset=orc.lib().orc_bfe_generator())
    co = np.zeros((32, 3), np.uint64)
    co[:8] = orc.random_elements(rng, (8, 3))
    cw = orc.coset_evaluate(co, d, fk=3)
    ch = orc.random_elements(rng, 3)
    folded = orc.fri_split_and_fold(cw, d, ch)
    want = np.zeros((16, 3), np.uint64)
    for j in range(4):
        want[j] = orc.xfe_add(co[2 * j], orc.xfe_mul(ch, co[2 * j + 1]))
    assert (folded.reshape(-1) == orc.coset_evaluate(want, orc.domain_pow(d, 2), fk=3)).all()


def test_deep_and_ood_consistency(orc):
    """DEEP quotient of a polynomial codeword by its true out-of-domain value is a polynomial of
    degree one less (stark.rs:2096-2103); out_of_domain_row equals direct evaluation of the
    randomized interpolants (master_table.rs:348-390)."""
    rng = np.random.default_rng(8)
    n, h, ncols = 16, 3, 2
    trace = orc.random_elements(rng, (ncols, n))
    rnd = orc.random_elements(rng, (ncols, h))
    pt = orc.random_elements(rng, 3)
    row = orc.out_of_domain_row(trace, rnd, pt)
    for c in range(ncols):
        poly = orc.randomized_column_interpolant(trace[c], rnd[c])
        lifted = np.zeros((2 * n, 3), np.uint64)
        lifted[:, 0] = poly
        assert (row[c] == orc.poly_eval_xfe(lifted, pt)).all()
    w = orc.random_elements(rng, (ncols, 3))
    comb = orc.weighted_sum_of_columns(trace, rnd, w)
    want = np.zeros(3, np.uint64)
    for c in range(ncols):
        want = orc.xfe_add(want, orc.xfe_mul(row[c], w[c]))
    assert (orc.poly_eval_xfe(comb, pt) == want).all()
    d = orc.domain_of_length(64, offset=orc.lib().orc_bfe_generator())
    cw = orc.coset_evaluate(comb, d, fk=3)
    deep = orc.deep_codeword(cw, d, pt, want)
    co = orc.coset_interpolate(deep, d, fk=3).reshape(64, 3)
    assert (co[n + h - 1:] == 0).all() and co[n + h - 2].any()


# ---- multi-block overwrite-mode hash_varlen, pinned by the reference's program digests ------------------------
def _program_digest(orc, source):
    from oracle.vm import isa

    words = isa.parse(source).to_bwords()
    return [int(v) for v in orc.from_mont(orc.hash_varlen(orc.to_mont(words)))], len(words)


def test_hash_simple_program(orc):
    """`triton_program!(halt).hash()` (/root/reference/triton-isa/src/program.rs:496-510)."""
    digest, _ = _program_digest(orc, "halt")
    assert digest == [0x4338de79520b3949, 0xe6a2129b28850dc9, 0xfd3cd0986a860450, 0x69fdba910ceba7bc, 0x7e5b118c9594c062]


def test_program_hash_is_unchanged(orc):
    """`program_executing_every_instruction().program.hash()` (/root/reference/triton-vm/src/stark.rs:4828-4838; program
    at :4639-4768): Program::hash = Tip5::hash_varlen(to_bwords()) (triton-isa/src/program.rs:399-402) over 295 words =
    30 absorb blocks.  The only reference-held pin of multi-block, overwrite-mode hash_varlen and of its padding."""
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "program_every_instruction.tasm")) as f:
        digest, n_words = _program_digest(orc, f.read())
    assert n_words > 10 * 20
    assert digest == [16104359835754349618, 14381287807966156775, 14760563195542097310, 2080121037799184588,
                      13105746022149139394]
