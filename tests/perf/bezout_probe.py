"""Time tvm_bezout_coefficients (csrc/bezout.hip) for 2^k distinct RAM pointers on the GPU and check the defining identity
a * rp + b * rp' = 1 at one random point.   usage: python tests/perf/bezout_probe.py [log2 ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402  (checker only)
from triton_vm_amd import Context  # noqa: E402

P = 2**64 - 2**32 + 1
ctx = Context(device=0)
for log_n in [int(a) for a in sys.argv[1:]] or [12, 16, 18, 20]:
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    roots = np.unique(np.concatenate([np.arange(n // 2, dtype=np.uint64), rng.integers(1 << 40, P, n, dtype=np.uint64)]))[:n]
    d_roots = ctx.to_device(orc.to_mont(roots))
    d_a, d_b = ctx.alloc(n), ctx.alloc(n)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        ctx._check(ctx.lib.tvm_bezout_coefficients(ctx.handle, d_roots.ptr, n, d_a.ptr, d_b.ptr), "tvm_bezout_coefficients")
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    a, b = orc.from_mont(d_a.download()[:n]), orc.from_mont(d_b.download()[:n])
    x = int(rng.integers(0, P, dtype=np.uint64))
    horner = lambda co: __import__("functools").reduce(lambda acc, v: (acc * x + int(v)) % P, co[::-1], 0)
    rp, fd = 1, 0
    for r in roots:
        r = int(r)
        fd = (fd * (x - r) + rp) % P
        rp = rp * (x - r) % P
    ok = (horner(a) * rp + horner(b) * fd) % P == 1
    print(f"2^{log_n} pointers: {best:.2f} ms (best of 3, synchronous call)  identity {'holds' if ok else 'FAILS'}", flush=True)
    assert ok
