"""oracle/tvm_oracle_fast.c -- the optimised CPU restatement that bench.py's `cpu_baseline` leg times -- against the textbook
oracle (oracle/tvm_oracle.c, itself pinned to the reference's vectors by tests/test_oracle_pins.py): every function bit for bit."""
import numpy as np
import pytest


def test_tip5_permutation_and_hashing(orc):
    rng = np.random.default_rng(1)
    for _ in range(50):
        st = orc.random_elements(rng, 16)
        assert (orc.fast.tip5_permutation(st) == orc.tip5_permutation(st)).all()
    edge = np.array([0, 1, orc.P - 1, 2**32 - 1, 2**32, 2**63, orc.P - 2**32] + [0] * 9, np.uint64)
    assert (orc.fast.tip5_permutation(edge) == orc.tip5_permutation(edge)).all()
    for width in (1, 9, 10, 11, 64, 379):
        rows = orc.random_elements(rng, (37, width))
        digests = orc.fast.hash_rows(rows)
        assert (digests == orc.hash_rows(rows)).all()
    leaves = orc.random_elements(rng, (64, 5))
    assert (orc.fast.merkle_tree(leaves) == orc.merkle_tree(leaves)).all()


@pytest.mark.parametrize("log_n,n_cols,h,expansion", [(4, 3, 5, 8), (6, 5, 17, 4), (3, 2, 8, 8)])
def test_lde_table(orc, log_n, n_cols, h, expansion):
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    trace, rnd = orc.random_elements(rng, (n_cols, n)), orc.random_elements(rng, (n_cols, h))
    ev = orc.domain_of_length(expansion * n, offset=orc.lib().orc_bfe_generator())
    assert (orc.fast.lde_table(trace, rnd, ev) == orc.lde_table(trace, rnd, ev, 1)).all()


def test_air_and_deep(orc):
    rng = np.random.default_rng(3)
    n, q = 8, 32
    g = orc.lib().orc_bfe_generator()
    main_rows, aux_rows = orc.random_elements(rng, (q, 379)), orc.random_elements(rng, (q, 91, 3))
    ch, w = orc.random_elements(rng, (63, 3)), orc.random_elements(rng, (604, 3))
    args = (main_rows, aux_rows, orc.domain_of_length(n), orc.domain_of_length(q, offset=g), ch, w)
    assert (orc.fast.quotients_combined(*args) == orc.quotients_combined(*args)).all()
    d = orc.domain_of_length(1 << 11, offset=g)
    cw, pt, val = orc.random_elements(rng, (d.length, 3)), orc.random_elements(rng, 3), orc.random_elements(rng, 3)
    assert (orc.fast.deep_codeword(cw, d, pt, val) == orc.deep_codeword(cw, d, pt, val)).all()
