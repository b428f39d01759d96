"""The pinned staging ring behind every small host array the library takes (csrc/ntt.hip: h2d_small): the caller's array is free
when the entry point returns, the copy is stream-ordered, and the ring wraps (one stream synchronisation) without losing a byte.
tvm_gather_elements stages its index list through the ring: index lists of ~0.9 MB (below the ring's quarter, above which the plain
copy-and-wait path is taken) walk a 4 MB ring around twice in ten calls; the caller's array is overwritten right after each call."""
import numpy as np


def test_ring_wraps_and_caller_arrays_are_free_on_return(ctx):
    rng = np.random.default_rng(5)
    n_src = 1 << 12
    src = rng.integers(0, 1 << 63, n_src, dtype=np.uint64)
    d_src = ctx.to_device(src)
    n_idx = 115_000                                     # 0.92 MB of indices per call
    for call in range(10):
        idx = rng.integers(0, n_src, n_idx, dtype=np.uint64)
        want = src[idx]
        scribble = idx.copy()
        out = np.zeros(n_idx, np.uint64)
        ctx._check(ctx.lib.tvm_gather_elements(ctx.handle, d_src.ptr, 1, scribble.ctypes.data, n_idx, out.ctypes.data), "tvm_gather_elements")
        scribble[:] = 0                                 # the index array was the caller's: free again
        assert (out == want).all(), f"call {call}"


def test_arrays_above_the_ring_quarter_take_the_plain_path(ctx):
    rng = np.random.default_rng(6)
    src = rng.integers(0, 1 << 63, 1 << 10, dtype=np.uint64)
    d_src = ctx.to_device(src)
    idx = rng.integers(0, 1 << 10, 200_000, dtype=np.uint64)   # 1.6 MB > a quarter of the ring
    out = np.zeros(idx.size, np.uint64)
    ctx._check(ctx.lib.tvm_gather_elements(ctx.handle, d_src.ptr, 1, idx.ctypes.data, idx.size, out.ctypes.data), "tvm_gather_elements")
    assert (out == src[idx]).all()
