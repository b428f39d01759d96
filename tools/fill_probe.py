#!/usr/bin/env python3
"""Time the degree-lowering fill at a given size on the GPU: python tools/fill_probe.py [log2_rows]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (before the Context: see triton_vm_amd/sharded.py)

from triton_vm_amd import Context  # noqa: E402
from triton_vm_amd import degree_lowering as dl  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
ctx = Context(0)
d_main = ctx.synthetic(379 * n, 1)
d_aux = ctx.synthetic(91 * n * 3, 2)
ch = np.random.default_rng(1).integers(0, 2**63, size=(63, 3), dtype=np.uint64)
for rep in range(3):
    ctx.timer_start()
    dl.fill_derived_main_columns(ctx, d_main, n)
    ms_main = ctx.timer_stop()
    ctx.timer_start()
    dl.fill_derived_aux_columns(ctx, d_main, d_aux, n, ch)
    ms_aux = ctx.timer_stop()
    # bytes the fill has to move: read the 149 + 49*3 given columns (and successor rows from cache), write 230 + 41*3
    moved = n * 8 * (149 + 230 + (379 + 49 * 3 + 41 * 3))
    print(json.dumps({"log2_rows": log_n, "rep": rep, "main_ms": round(ms_main, 3), "aux_ms": round(ms_aux, 3),
                      "GB_per_s": round(moved / ((ms_main + ms_aux) * 1e-3) / 1e9, 1)}))
