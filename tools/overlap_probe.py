"""Does row hashing (VALU-bound, 32 VGPRs, 3.4 KB LDS) overlap with the LDE kernels (LDS-bound occupancy, VALU ~60 % busy)
when both run at once on two streams of one GPU?  Two contexts (= two streams): context A extends a main table, context B
hashes the rows of an already extended one.  Prints serial vs concurrent wall time.
usage: python tools/overlap_probe.py [log2_rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triton_vm_amd import Context  # noqa: E402
from triton_vm_amd.prover import Prover, StarkParameters  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
a, b = Context(device=0), Context(device=0)
p = StarkParameters(log_n)
pa, pb = Prover(a, p, seed=1), Prover(b, p, seed=2)
pa.aux.d_trace.free(); pb.aux.d_trace.free()
L = p.ldt.length
pb.main.maybe_low_degree_extend_all_columns()
d_digests = b.alloc(5 * L)


def lde():
    pa.main.maybe_low_degree_extend_all_columns()


def hash_rows():
    b._check(b.lib.tvm_hash_rows(b.handle, pb.main._need_table(), L, d_digests.ptr), "hash")


def wall(fns):
    a.sync(); b.sync()
    t = time.perf_counter()
    for f in fns:
        f()
    a.sync(); b.sync()
    return 1e3 * (time.perf_counter() - t)


for f in (lde, hash_rows):
    wall([f])
t_lde = min(wall([lde]) for _ in range(3))
t_hash = min(wall([hash_rows]) for _ in range(3))
t_both = min(wall([lde, hash_rows]) for _ in range(3))
t_both2 = min(wall([hash_rows, lde]) for _ in range(3))
print(f"2^{log_n} rows: LDE {t_lde:.1f} ms, row hashing {t_hash:.1f} ms, serial {t_lde + t_hash:.1f} ms; "
      f"concurrent {t_both:.1f} ms (LDE issued first), {t_both2:.1f} ms (hashing issued first)")
