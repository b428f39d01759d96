#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a per-kernel table (like --stats).
Usage: python tools/rocprof_summary.py <results.db> [> profiles/NAME.txt]"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(workgroup_x), max(grid_x), max(grid_y), max(grid_z) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"{'kernel':58s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} "
          f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'wg':>5s}")
    for r in rows:
        print(f"{r[0][:58]:58s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} {r[5] / 1e3:10.1f} "
              f"{100 * r[2] / total:6.1f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:7d} {r[11]:5d}")


if __name__ == "__main__":
    main(sys.argv[1])
