"""Time tvm_extend_aux_table and the two degree-lowering fills at 2^k rows on the GPU (valid 2048-row trace tiled).
usage: python tests/perf/extend_probe.py [log2_rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from tests import vm_fixture as vf  # noqa: E402
from triton_vm_amd import Context, degree_lowering as dl  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = Context(device=0)
main, aux, ch, _ = vf.valid_tables("every")
n = 1 << log_n
big = np.ascontiguousarray(np.tile(main, (1, n // main.shape[1])))
d_main = ctx.to_device(big)
d_aux = ctx.to_device(np.zeros((91, n, 3), np.uint64))
for name, fn in (("extend (49 columns)", lambda: ctx._check(ctx.lib.tvm_extend_aux_table(ctx.handle, d_main.ptr, d_aux.ptr, n, ch.ctypes.data), "extend")),
                 ("fill derived aux", lambda: dl.fill_derived_aux_columns(ctx, d_main, d_aux, n, ch)),
                 ("fill derived main", lambda: dl.fill_derived_main_columns(ctx, d_main, n))):
    fn()
    ms = []
    for _ in range(5):
        ctx.timer_start()
        fn()
        ms.append(ctx.timer_stop())
    print(f"{name}: {min(ms):.3f} ms (min of 5) at 2^{log_n} rows")
# algorithmic bytes of extend: read 149 main columns once, write 49 XFE columns once
b = n * (149 * 8 + 49 * 24)
print(f"extend algorithmic bytes {b / 1e9:.2f} GB")
