"""The context's side lane (include/triton_hip.h: tvm_side_*): a second stream ordered against the context's stream by events only,
on which the asynchronous coefficient exchange of the column split runs (triton_vm_amd/host: rccl_comm.cpp, sharded_host.cpp).
What is checked here, through the C ABI on the emulation and on the MI355X: a copy on the side lane sees everything queued on the
context's stream before tvm_side_begin; work queued on the context's stream after tvm_side_wait sees the copy; slots are independent;
a slot that was never marked is no wait; bad slots are refused."""
import numpy as np


def test_copies_on_the_side_lane_are_ordered_by_begin_mark_and_wait(ctx, orc):
    lib, h = ctx.lib, ctx.handle
    n = 1 << 16
    a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    assert lib.tvm_ctx_side_stream(h) or ctx.kind == "emu"   # (the emulation's second stream is a token)
    for seed in (11, 12, 13):
        # producer on the context's stream -> copy on the side lane -> consumer on the context's stream
        ctx._check(lib.tvm_synthetic_fill(h, a.ptr, n, seed), "fill")
        ctx._check(lib.tvm_side_begin(h), "side_begin")
        ctx._check(lib.tvm_side_memcpy_d2d(h, b.ptr, a.ptr, 8 * n), "side_memcpy")
        ctx._check(lib.tvm_side_mark(h, seed % 16), "side_mark")
        ctx._check(lib.tvm_side_wait(h, seed % 16), "side_wait")
        ctx._check(lib.tvm_memcpy_d2d(h, c.ptr, b.ptr, 8 * n), "consumer")
        want = a.download()
        assert (c.download() == want).all() and want.any()
    ctx._check(lib.tvm_side_sync(h), "side_sync")
    ctx._check(lib.tvm_side_wait(h, 7), "a slot that was never marked is no wait")
    assert lib.tvm_side_mark(h, 16) != 0 and lib.tvm_side_wait(h, 99) != 0   # TVM_SIDE_SLOTS = 16
    assert lib.tvm_side_memcpy_d2d(h, None, a.ptr, 8) != 0


def test_two_exchanges_in_flight_under_work_on_the_contexts_stream(ctx, orc):
    """the shape of MasterTable::low_degree_extend_over: request k + 1, wait k, consume k -- two slots alternating"""
    lib, h = ctx.lib, ctx.handle
    n, chunks = 1 << 14, 5
    src = [ctx.alloc(n) for _ in range(chunks)]
    dst = [ctx.alloc(n) for _ in range(chunks)]
    out = ctx.alloc(n)
    for k in range(chunks):
        ctx._check(lib.tvm_synthetic_fill(h, src[k].ptr, n, 100 + k), "fill")

    def request(k):
        ctx._check(lib.tvm_side_begin(h), "side_begin")
        ctx._check(lib.tvm_side_memcpy_d2d(h, dst[k].ptr, src[k].ptr, 8 * n), "side_memcpy")
        ctx._check(lib.tvm_side_mark(h, k % 2), "side_mark")

    request(0)
    for k in range(chunks):
        if k + 1 < chunks:
            request(k + 1)
        ctx._check(lib.tvm_side_wait(h, k % 2), "side_wait")
        ctx._check(lib.tvm_memcpy_d2d(h, out.ptr, dst[k].ptr, 8 * n), "consumer")
        assert (out.download() == src[k].download()).all()
