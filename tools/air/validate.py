"""Reproduce the reference's own constraint-overview statistics
(/root/reference/triton-vm/src/table/master_table.rs:1831-2040, whose output is committed at
/root/reference/specification/src/arithmetization-overview.md:24-81) with this port: per-table builders,
per-table degree lowering, structurally-unique node counts.  Every number in that file must match."""
from . import build
from .circuit import Builder, lower_to_degree, multicircuit_degree, reachable
from .defs import AUX_START, MAIN_START
from . import names

EXPECTED = {
    None: {"counts": [(6, 5, 10, 2), (29, 10, 42, 1), (3, 1, 5, 0), (7, 1, 12, 1), (6, 0, 6, 0), (22, 45, 48, 2),
                      (2, 1, 3, 0), (3, 1, 4, 1), (1, 15, 22, 2), (0, 0, 0, 14)],
           "nodes": (539, 637, 6825, 213), "max_degree": [4, 19, 4, 5, 5, 9, 4, 3, 12, 1]},
    8: {"counts": [(6, 5, 10, 2), (29, 10, 149, 1), (3, 1, 5, 0), (7, 1, 12, 1), (6, 0, 6, 0), (22, 46, 50, 2),
                   (2, 1, 3, 0), (3, 1, 4, 1), (1, 18, 24, 2), (0, 0, 0, 14)],
        "nodes": (539, 648, 7059, 213)},
    4: {"counts": [(6, 5, 10, 2), (31, 10, 242, 1), (3, 1, 5, 0), (7, 1, 13, 1), (6, 0, 7, 0), (22, 52, 85, 2),
                   (2, 1, 3, 0), (3, 1, 4, 1), (1, 26, 34, 2), (0, 0, 0, 14)],
        "nodes": (543, 689, 7400, 213)},
}


def _table_ends():
    ends, m, a = {}, 0, 0
    for t in names.TABLES:
        m = MAIN_START[t] + len(names.MAIN_COLUMNS[t])
        a = AUX_START[t] + len(names.AUX_COLUMNS[t])
        ends[t] = (m, a)
    ends["GrandCrossTableArg"] = (0, 0)
    return ends


def structural_unique_count(root_nodes):
    intern, memo = {}, {}
    for n in reachable(root_nodes):
        if n.kind == "op":
            key = ("op", n.op, memo[id(n.lhs)], memo[id(n.rhs)])
        else:
            key = (n.kind, n.val)
        memo[id(n)] = intern.setdefault(key, len(intern))
    return len(intern)


def overview(target_degree):
    ends = _table_ends()
    counts, all_roots, max_degrees = [], {s: [] for s, _, _ in build.SECTIONS}, []
    for name, mod in build.TABLES:
        row, degs = [], []
        for sec, fn, dual in build.SECTIONS:
            b = Builder(dual=dual)
            roots = getattr(mod, fn)(b)
            kept = [r.node for r in roots]               # the test's `.clone()` before lowering
            if target_degree is not None:
                new_main, new_aux = lower_to_degree(roots, b, target_degree, ends[name][0], ends[name][1])
                kept += [m.node for m in new_main] + [m.node for m in new_aux]
            row.append(len(kept))
            degs.append(multicircuit_degree(kept) if kept else -1)
            all_roots[sec] += kept
        counts.append(tuple(row))
        max_degrees.append(max(degs))
    nodes = tuple(structural_unique_count(all_roots[s]) for s, _, _ in build.SECTIONS)
    return counts, nodes, max_degrees


def main():
    ok = True
    for target, want in EXPECTED.items():
        counts, nodes, max_degrees = overview(target)
        good = counts == want["counts"] and nodes == want["nodes"] and (
            "max_degree" not in want or max_degrees == want["max_degree"])
        ok &= good
        print(f"target degree {target}: counts {'ok' if counts == want['counts'] else counts}, "
              f"nodes {nodes} (want {want['nodes']}) -> {'MATCH' if good else 'MISMATCH'}")
        if "max_degree" in want and max_degrees != want["max_degree"]:
            print("   max degrees", max_degrees, "want", want["max_degree"])
    return ok


if __name__ == "__main__":
    raise SystemExit(0 if main() else 1)
