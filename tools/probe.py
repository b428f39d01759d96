#!/usr/bin/env python3
"""GPU probe: time the hot-path stages that exist, at a given size, with HIP events.
Usage: python tools/probe.py [log_n] [n_main_cols] [n_aux_cols]  -> JSON lines on stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

from triton_vm_amd import ArithmeticDomain, Context, field  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n_main = int(sys.argv[2]) if len(sys.argv) > 2 else 379
    n_aux = int(sys.argv[3]) if len(sys.argv) > 3 else 91
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    n, h = 1 << log_n, 198
    variant = os.environ.get('TVM_LIB_VARIANT')  # experiment libraries built with build(variant=...)
    if variant:
        from triton_vm_amd.capi import load_library
        ctx = Context(0, lib=load_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'triton_vm_amd', f'libtriton_hip_{variant}.so')))
    else:
        ctx = Context(0)
    for opt, val in [kv.split("=") for kv in os.environ.get("TVM_PROBE_OPTIONS", "").split(",") if kv]:   # e.g. "2=32": TVM_OPTION_LDE_CHUNK_COLUMNS
        ctx._check(ctx.lib.tvm_ctx_set_option(ctx.handle, int(opt), int(val)), "tvm_ctx_set_option")
    trace_dom = ArithmeticDomain.of_length(n)
    expansion = int(os.environ.get("TVM_PROBE_EXPANSION", "8"))   # |evaluation domain| / |trace domain| (32: FRI log-blowup 4)
    ev = ArithmeticDomain.of_length(expansion * n).with_offset(field.generator())
    L = len(ev)
    for name, fk, n_cols in (("main", 1, n_main), ("aux", 3, n_aux)):
        if not n_cols:
            continue
        d_trace = ctx.synthetic(n_cols * n * fk, 1)
        d_rnd = ctx.synthetic(n_cols * h * fk, 2)
        d_nodes = ctx.alloc(10 * L)
        cells = n * n_cols * fk
        for rep in range(reps):
            t = C.c_void_p()
            ctx.timer_start()
            ctx._check(ctx.lib.tvm_lde_table(ctx.handle, fk, d_trace.ptr, n, n_cols, d_rnd.ptr, h, trace_dom.c(), ev.c(),
                                             C.byref(t)), "lde")
            ms_lde = ctx.timer_stop()
            ctx.timer_start()
            ctx._check(ctx.lib.tvm_hash_rows(ctx.handle, t, L, d_nodes.ptr + 40 * L), "hash")
            ms_hash = ctx.timer_stop()
            ctx.timer_start()
            ctx._check(ctx.lib.tvm_merkle_tree(ctx.handle, d_nodes.ptr + 40 * L, L, d_nodes.ptr), "merkle")
            ms_merkle = ctx.timer_stop()
            ctx.lib.tvm_table_free(ctx.handle, t)
            print(json.dumps({"table": name, "log_n": log_n, "cols": n_cols, "rep": rep, "lde_ms": round(ms_lde, 3),
                              "hash_ms": round(ms_hash, 3), "merkle_ms": round(ms_merkle, 3),
                              "lde_GBps_algorithmic": round(cells * (8 + 8 * expansion) / ms_lde / 1e6, 1),
                              "hash_Mperm_per_s": round(L * (n_cols * fk // 10 + 1) / ms_hash / 1e3, 1)}), flush=True)
        d_trace.free(); d_rnd.free(); d_nodes.free()
    ctx.close()


if __name__ == "__main__":
    main()
