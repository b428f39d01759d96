"""Time the host `gen` tail on the device at BASELINE size: tvm_fill_main_table -> tvm_pad_main_table ->
tvm_fill_derived_main_columns -> tvm_extend_aux_table -> tvm_fill_derived_aux_columns, on an AET made by repeating the
trace arrays of `program_executing_every_instruction` until the processor table has ~2^k rows (timing only: a repeated
trace is not a valid execution).  usage: python tests/perf/gen_tail_probe.py [log2_rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from tests import vm_fixture as vf  # noqa: E402
from tests.test_fill import aet_arrays  # noqa: E402
from triton_vm_amd import Context, degree_lowering as dl, master_table as mtab  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
_, aet, _, _ = vf.run("every")
a = aet_arrays(orc, aet)
reps = (n - 8) // max(a[k].shape[0] for k in ("processor_trace", "op_stack_trace", "ram_trace"))
big = dict(a)
for key in ("processor_trace", "op_stack_trace", "ram_trace", "sponge_trace", "hash_trace", "u32_entries"):
    big[key] = np.ascontiguousarray(np.tile(a[key], (reps if key != "u32_entries" else min(reps, 40), 1)))
clk = np.arange(big["processor_trace"].shape[0], dtype=object)          # a running clock keeps the histogram in range
big["processor_trace"][:, 0] = orc.to_mont(clk)
h = min(big["sponge_trace"].shape[0] + big["hash_trace"].shape[0], n - 256) // 2
big["sponge_trace"], big["hash_trace"] = big["sponge_trace"][:h], big["hash_trace"][:h]
ctx = Context(device=0)
d_main = ctx.alloc(379 * n)
d_aux = ctx.alloc(91 * n * 3)
ch = orc.random_elements(np.random.default_rng(1), (63, 3))
bytes_in = sum(v.nbytes for v in big.values())


def timed(name, fn, reps_=3):
    ms = []
    for _ in range(reps_):
        ctx.sync()
        t0 = time.perf_counter()
        out = fn()
        ctx.sync()
        ms.append(1e3 * (time.perf_counter() - t0))
    print(f"{name}: {min(ms):.2f} ms wall (min of {reps_}) at 2^{log_n} rows")
    return out


lengths = timed(f"fill from the AET ({bytes_in / 1e9:.2f} GB of host arrays, incl. the upload)", lambda: mtab.fill(ctx, d_main, n, big))
timed("pad", lambda: ctx._check(ctx.lib.tvm_pad_main_table(ctx.handle, d_main.ptr, n, np.array(lengths, np.uint64).ctypes.data), "pad"))
timed("fill derived main", lambda: dl.fill_derived_main_columns(ctx, d_main, n))
timed("extend", lambda: ctx._check(ctx.lib.tvm_extend_aux_table(ctx.handle, d_main.ptr, d_aux.ptr, n, ch.ctypes.data), "extend"))
timed("fill derived aux", lambda: dl.fill_derived_aux_columns(ctx, d_main, d_aux, n, ch))
print("table lengths", lengths)
