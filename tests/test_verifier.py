"""The verifier's batch work over the revealed rows (csrc/verify.hip): leaf digests against the oracle's
hash_varlen; the per-row linear combination + DEEP value against a python-integer restatement of stark.rs:1678-1755;
and end to end -- the values recomputed from the rows a device PROOF opened must equal the prover's own combination
codeword at the queried indices (the equality Verifier::verify checks at stark.rs:1749-1752)."""
import numpy as np
import pytest

from triton_vm_amd import ArithmeticDomain, field, verifier
from triton_vm_amd.prover import Prover, StarkParameters

P = 2**64 - 2**32 + 1


def _xmul(a, b):
    c0, c1, c2 = a[0] * b[0], a[0] * b[1] + a[1] * b[0], a[0] * b[2] + a[1] * b[1] + a[2] * b[0]
    c3, c4 = a[1] * b[2] + a[2] * b[1], a[2] * b[2]
    return [(c0 - c3) % P, (c1 + c3 - c4) % P, (c2 + c4) % P]


def _python_deep_value(orc, main_row, aux_row, quot_row, x, wma, wq, wd, pts, vals):
    f = lambda a: [[int(v) for v in e] for e in orc.from_mont(np.asarray(a, np.uint64).reshape(-1, 3))]
    wma, wq, wd, pts, vals, aux_row, quot_row = map(f, (wma, wq, wd, pts, vals, aux_row, quot_row))
    main_row = [int(v) for v in orc.from_mont(np.asarray(main_row, np.uint64))]
    add = lambda a, b: [(s + t) % P for s, t in zip(a, b)]
    ma = [0, 0, 0]
    for c in range(379):
        ma = add(ma, [w * main_row[c] % P for w in wma[c]])
    for c in range(91):
        ma = add(ma, _xmul(wma[379 + c], aux_row[c]))
    shared = [0, 0, 0]
    for k in (1, 2, 3):
        shared = add(shared, _xmul(quot_row[k], wq[k]))
    for_p = add(_xmul(wq[0], quot_row[0]), shared)
    for_r = add(_xmul(wq[4], quot_row[4]), shared)
    total = [0, 0, 0]
    for k, elem in enumerate((ma, ma, for_p, for_r)):
        num = [(s - t) % P for s, t in zip(elem, vals[k])]
        den = [(x - pts[k][0]) % P, (-pts[k][1]) % P, (-pts[k][2]) % P]
        inv = [int(v) for v in orc.from_mont(orc.xfe_inv(orc.to_mont(den)))]
        total = add(total, _xmul(wd[k], _xmul(num, inv)))
    return total


@pytest.mark.parametrize("n", [1, 5, 40])
def test_row_digests_and_deep_values_match_restatement(ctx, orc, n):
    rng = np.random.default_rng(n)
    r = lambda *s: orc.random_elements(rng, s)
    main_rows, aux_rows, quot_rows = r(n, 379), r(n, 91, 3), r(n, 5, 3)
    for rows in (main_rows, aux_rows, quot_rows):
        assert (verifier.row_digests(ctx, rows) == orc.hash_rows(rows.reshape(n, -1))).all()
    ldt = ArithmeticDomain.of_length(1 << 12).with_offset(field.generator())
    idx = rng.integers(0, len(ldt), n).astype(np.uint64)
    wma, wq, wd, pts, vals = r(470, 3), r(5, 3), r(4, 3), r(4, 3), r(4, 3)
    got = verifier.deep_values(ctx, main_rows, aux_rows, quot_rows, idx, ldt, wma, wq, wd, pts, vals)
    gen, off = orc.value(ldt.generator), orc.value(ldt.offset)
    for j in range(n):
        x = off * pow(gen, int(idx[j]), P) % P
        want = _python_deep_value(orc, main_rows[j], aux_rows[j], quot_rows[j], x, wma, wq, wd, pts, vals)
        assert [int(v) for v in orc.from_mont(got[j])] == want, j


def _check_proof_rows(ctx, orc, p, prover):
    c = prover.capture
    # the items the verifier reads (stark.rs:1386-1600), straight from the captured transcript data
    alpha = c["alpha"]
    lib = ctx.lib
    from triton_vm_amd.prover import xfe_add, xfe_mul, xfe_powers

    alpha_next = np.array([field.mont_mul(int(v), p.trace.generator) for v in alpha], np.uint64)
    a4 = xfe_powers(lib, alpha, 4, 1)[0]
    za4 = xfe_powers(lib, np.array([field.mont_mul(int(v), field.to_mont(3)) for v in alpha], np.uint64), 4, 1)[0]
    wma, wq, wd = c["weights_ma"], c["weights_q"], c["weights_d"]

    def row_sum(main_row, aux_row):                      # out-of-domain rows -> their linear combination
        acc = np.zeros(3, np.uint64)
        for k in range(379):
            acc = xfe_add(acc, xfe_mul(lib, wma[k], main_row[k]))
        for k in range(91):
            acc = xfe_add(acc, xfe_mul(lib, wma[379 + k], aux_row[k]))
        return acc

    seg = c["seg_ood"]                                   # [5][2][3]: segment k at alpha^4 and at (zeta alpha)^4
    p_val, r_val = np.zeros(3, np.uint64), np.zeros(3, np.uint64)
    for k in range(4):
        p_val = xfe_add(p_val, xfe_mul(lib, wq[k], seg[k, 0]))
    for k in range(1, 5):
        r_val = xfe_add(r_val, xfe_mul(lib, wq[k], seg[k, 1]))
    ood_values = [row_sum(c["ood_main"][0], c["ood_aux"][0]), row_sum(c["ood_main"][1], c["ood_aux"][1]), p_val, r_val]
    opened_at = np.array(prover.opened_at, np.uint64)
    quot_rows = np.asarray(prover.opened_quotient_rows, np.uint64)
    got = verifier.deep_values(ctx, prover.opened["main"], prover.opened["aux"], quot_rows, opened_at, p.ldt, wma, wq, wd,
                               [alpha, alpha_next, a4, za4], ood_values)
    assert (got == c["combination"][opened_at.astype(np.int64)]).all()


def test_values_recomputed_from_a_proofs_opened_rows_equal_the_combination_codeword(ctx, orc):
    rng = np.random.default_rng(12)
    p = StarkParameters(3, num_trace_randomizers=3, num_collinearity_checks=4)
    prover = Prover(ctx, p, orc.random_elements(rng, (379, p.trace.length)), orc.random_elements(rng, (91, p.trace.length, 3)), seed=5)
    prover.capture = {}
    prover.prove()
    _check_proof_rows(ctx, orc, p, prover)


@pytest.mark.gpu
def test_full_size_proof_rows_against_the_combination_codeword(orc):
    """2^20 padded rows: the 173 rows a proof opens, pushed through the verifier-side kernels, reproduce the prover's
    combination codeword at the queried indices -- AIR-independent end-to-end check of linear combinations, out-of-domain
    rows, quotient segments, DEEP and the row openings at the BASELINE size."""
    from triton_vm_amd import Context

    ctx = Context(0)
    p = StarkParameters(20)
    prover = Prover(ctx, p, seed=4)
    prover.capture = {}
    prover.prove()
    _check_proof_rows(ctx, orc, p, prover)
    ctx.close()
