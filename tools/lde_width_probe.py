"""EXPERIMENT: main-table LDE time per column as a function of the table's width (the width sets how far apart the 1024
per-transform-index regions of pass 3's stores lie: 8192 rows x W x 8 bytes).  usage: python tools/lde_width_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from triton_vm_amd import ArithmeticDomain, Context, field  # noqa: E402
from triton_vm_amd.master_table import MasterTable  # noqa: E402

ctx = Context(device=0)
n, H = 1 << 20, 198
trace_dom = ArithmeticDomain.of_length(n)
ev = ArithmeticDomain.of_length(8 * n).with_offset(field.generator())
for n_cols in (96, 192, 379, 96):
    mt = MasterTable.from_device(ctx, ctx.synthetic(n_cols * n, seed=3), ctx.synthetic(n_cols * H, seed=4), n_cols, n, H, trace_dom, ev, ev, 1)
    ms = []
    for _ in range(4):
        mt.clear_cache()
        ctx.timer_start()
        mt.maybe_low_degree_extend_all_columns()
        ms.append(ctx.timer_stop())
    mt.clear_cache()
    print(f"W = {n_cols}: {min(ms):.2f} ms, {1e3 * min(ms) / n_cols:.1f} us per column (regions {8192 * n_cols * 8 / 1e6:.1f} MB apart)", flush=True)
    del mt
