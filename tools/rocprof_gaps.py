#!/usr/bin/env python3
"""Where the device idles inside one proof: from a rocprofv3 (rocpd sqlite) kernel trace, take the LAST proof of the run
(from the last k_fa_* / k_fill_main_* kernel before the final batch to the last kernel), and list the largest gaps between
consecutive kernels with the kernels on either side.
Usage: python tools/rocprof_gaps.py <results.db> [first-kernel-substring]"""
import sqlite3
import sys


def main(path, first="k_pad_main_table"):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if not starts:
        print("no", first, "kernel in the trace")
        return
    # one proof: from the last marker kernel to the next marker (or the end)
    for which, i0 in enumerate(starts):
        i1 = starts[which + 1] if which + 1 < len(starts) else len(rows)
        seg = rows[i0:i1]
        span = (seg[-1][2] - seg[0][1]) / 1e6
        busy = sum(r[2] - r[1] for r in seg) / 1e6
        gaps = sorted(((seg[k + 1][1] - seg[k][2]) / 1e3, seg[k][0][:40], seg[k + 1][0][:40]) for k in range(len(seg) - 1))
        big = [g for g in gaps if g[0] > 15]
        print(f"segment {which}: {len(seg)} kernels, span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms; "
              f"{len(big)} gaps > 15 us summing to {sum(g[0] for g in big) / 1e3:.2f} ms")
        for g in sorted(big, reverse=True)[:14]:
            print(f"    {g[0]:9.1f} us  after {g[1]:40s} before {g[2]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
