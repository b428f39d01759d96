#!/usr/bin/env python3
"""Host side of one proof from a rocprofv3 --kernel-trace --hip-trace (rocpd sqlite) run: for a proof from the MIDDLE of the run (one of bench.py's timed steps; from
its first fill kernel to the next proof's), the HIP API calls that took longest, the totals per API,
and for every device-idle gap above a threshold the API calls the host was inside meanwhile.
Usage: python tools/rocprof_host_timeline.py <results.db> [gap_us] [file for the merged kernel / API timeline of that proof]"""
import sqlite3
import sys


def main(path, gap_us="40", dump=""):
    gap_ns = float(gap_us) * 1e3
    con = sqlite3.connect(path)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kernels = list(cur.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(kernels) if "k_fill_main_init" in r[0]]
    if not marks:
        print("no k_fill_main_init kernel in the trace")
        return
    # (bench.py's last proofs are the ones with the stage timers -- an event synchronisation per stage -- and the Python mirror
    # host's; a proof from the middle of the run is one of the timed steps of the C++ host)
    which = len(marks) // 2 if len(marks) >= 3 else len(marks) - 1
    seg = kernels[marks[which]:marks[which + 1]] if which + 1 < len(marks) else kernels[marks[which]:]
    t0, t1 = seg[0][1], seg[-1][2]
    view = "regions" if "regions" in names else None
    if view is None:
        print("no regions view; tables:", names)
        return
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    api = list(cur.execute(f"select name, start, end from {view} where start >= ? and start <= ? order by start", (t0 - 2_000_000, t1)))
    print(f"# {path}: last proof {len(seg)} kernels, device span {(t1 - t0) / 1e6:.3f} ms, {len(api)} host API regions (columns {cols})")
    tot = {}
    for n, s, e in api:
        d = tot.setdefault(n, [0, 0])
        d[0] += 1
        d[1] += e - s
    print("## host API totals inside the proof")
    for n, (k, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{n:40s} {k:6d} calls {ns / 1e6:9.3f} ms")
    if dump:   # the merged timeline of the proof: kernels (K) and host API regions (A) by start time, microseconds from the first kernel
        merged = [(s0, "K", n, e0 - s0) for n, s0, e0 in seg] + [(s0, "A", n, e0 - s0) for n, s0, e0 in api if s0 >= t0]
        with open(dump, "w") as f:
            for s0, kind, n, d in sorted(merged):
                f.write(f"{(s0 - t0) / 1e3:10.1f} {kind} {d / 1e3:8.1f} {n[:70]}\n")
    print(f"## device-idle gaps > {gap_us} us and what the host was in")
    for k in range(len(seg) - 1):
        g0, g1 = seg[k][2], seg[k + 1][1]
        if g1 - g0 < gap_ns:
            continue
        inside = [(n, s, e) for n, s, e in api if e > g0 and s < g1]
        inside.sort(key=lambda r: -(min(r[2], g1) - max(r[1], g0)))
        what = ", ".join(f"{n} {((min(e, g1) - max(s, g0)) / 1e3):.0f}us" for n, s, e in inside[:4])
        print(f"{(g0 - t0) / 1e6:8.3f} ms  gap {(g1 - g0) / 1e3:8.1f} us  after {seg[k][0][:36]:36s} before {seg[k + 1][0][:36]:36s} | {len(inside)} calls: {what}")


if __name__ == "__main__":
    main(*sys.argv[1:])
