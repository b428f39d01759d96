#!/usr/bin/env python3
"""Time all_quotients_combined alone at 2^20 rows (synthetic tables), optionally with an experiment
library:  python tools/air_bench.py [variant ...]   ->  one JSON line per library."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from triton_vm_amd import stark
from triton_vm_amd.capi import Context, load_library
from triton_vm_amd.prover import Prover, StarkParameters


def run(variant, log2_rows=20):
    path = None if variant == "default" else os.path.join(ROOT, "triton_vm_amd", f"libtriton_hip_{variant}.so")
    ctx = Context(device=0, lib=load_library(path))
    p = StarkParameters(log2_rows)
    prover = Prover(ctx, p, seed=7)
    prover.main.maybe_low_degree_extend_all_columns()
    prover.aux.maybe_low_degree_extend_all_columns()
    rng = np.random.default_rng(3)
    ch = rng.integers(0, 2**63, size=(63, 3), dtype=np.uint64)
    w = rng.integers(0, 2**63, size=(604, 3), dtype=np.uint64)
    if os.environ.get("AIR_VALID_TRACE") == "1":
        ctx.assume_valid_trace(True)
    ms = []
    for _ in range(3):
        ctx.timer_start()
        q = stark.all_quotients_combined(ctx, prover.main, prover.aux, p.trace, p.quotient, ch, w)
        ms.append(round(ctx.timer_stop(), 3))
        del q
    print(json.dumps({"variant": variant, "log2_rows": log2_rows, "air_ms": ms}), flush=True)
    ctx.close()


if __name__ == "__main__":
    for v in sys.argv[1:] or ["default"]:
        run(v)
