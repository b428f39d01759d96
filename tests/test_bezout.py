"""csrc/bezout.hip (tvm_bezout_coefficients: the RAM table's Bezout coefficient polynomials,
/root/reference/triton-vm/src/table/ram.rs:152-207) against the oracle's quadratic restatement
(oracle/vm/tables.py::bezout_coefficient_polynomials_coefficients) for small root sets, and against the defining identity
a * rp + b * rp' = 1 at random points for large ones (subproduct and remainder trees with several transform levels)."""
import numpy as np
import pytest

P = 2**64 - 2**32 + 1


def device_bezout(ctx, orc, roots):
    n = len(roots)
    d_roots = ctx.to_device(orc.to_mont(np.array(roots, dtype=object)) if n else np.zeros(1, np.uint64))
    d_a, d_b = ctx.alloc(max(n, 1)), ctx.alloc(max(n, 1))
    ctx._check(ctx.lib.tvm_bezout_coefficients(ctx.handle, d_roots.ptr, n, d_a.ptr, d_b.ptr), "tvm_bezout_coefficients")
    if not n:
        return [], []
    return [int(v) for v in orc.from_mont(d_a.download()[:n])], [int(v) for v in orc.from_mont(d_b.download()[:n])]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 63, 64, 65, 100, 129, 300])
def test_bezout_coefficients_match_the_oracle(ctx, orc, n):
    from oracle.vm import tables as T

    rng = np.random.default_rng(n)
    roots = [int(v) for v in dict.fromkeys(int(x) for x in rng.integers(0, P, n + 5, dtype=np.uint64))][:n]
    if n >= 5:
        roots[:5] = [0, 1, P - 1, 7, 2**32]   # small and special pointers, as real RAM pointers are
    want_a, want_b = T.bezout_coefficient_polynomials_coefficients(roots)
    got_a, got_b = device_bezout(ctx, orc, roots)
    assert got_b == want_b
    assert got_a == want_a


@pytest.mark.parametrize("n", [1000, 5000, 20000, 300000])
def test_bezout_identity_holds_for_large_root_sets(ctx, orc, n):
    if n > 20000 and ctx.kind == "emu":
        pytest.skip("CPU suite time: the largest set (2^19 padded leaves, 14 transform levels) runs on the GPU")
    rng = np.random.default_rng(n)
    roots = list(dict.fromkeys([int(v) for v in range(n // 2)] + [int(x) for x in rng.integers(0, P, n, dtype=np.uint64)]))[:n]
    a, b = device_bezout(ctx, orc, roots)
    assert a[n - 1] == 0      # deg a = n - 2
    horner = lambda coeffs, x: __import__("functools").reduce(lambda acc, c: (acc * x + c) % P, reversed(coeffs), 0)
    for x in [int(v) for v in rng.integers(0, P, 4, dtype=np.uint64)] + [roots[3]]:
        rp, fd = 1, 0
        for r in roots:                      # rp and rp' at x by the product rule
            fd = (fd * (x - r) + rp) % P
            rp = rp * (x - r) % P
        assert (horner(a, x) * rp + horner(b, x) * fd) % P == 1


def test_repeated_roots_are_refused(ctx, orc):
    from triton_vm_amd.capi import TritonHipError

    with pytest.raises(TritonHipError):
        device_bezout(ctx, orc, [5, 9, 5])
