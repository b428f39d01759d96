#!/usr/bin/env python3
"""Export the degree-lowering FILL: the values of the derived main / auxiliary columns.

The reference generates `DegreeLoweringTable::fill_derived_main_columns / fill_derived_aux_columns` at build time
(/root/reference/triton-constraint-builder/src/substitutions.rs:128-205, rules at :215-233, row loops at
:236-400): for every substitution  x - expr  that the degree lowering introduced, column x of every row is
expr evaluated on that row (single-row sections: all rows) or on the row and its successor (transition section:
rows 0 .. n-2).  A rule may read the columns derived before it in the same section of the same row, and any column
of the earlier sections.  The substitutions themselves come from tools/air/build.py (the same deterministic
lowering that yields the AIR the quotient kernels evaluate): 230 of the 379 main columns and 41 of the 90
auxiliary columns are derived.

  triton_vm_amd/csrc/fill_gen.hip   -- one work-item per row, straight-line code per section (generated, committed)
  oracle/degree_lowering_rules.py   -- the same rules as plain data for the CPU oracle (oracle/degree_lowering.py)

Regenerate with `python -m tools.air.export_fill`.
"""
import os

from . import build
from .circuit import evaluates_to_base_element
from .export import ROOT, Scheduler, lit, mont

SECTIONS = ("init", "cons", "tran", "term")
KIND = {"main": "MC", "cm": "MC", "nm": "MN", "aux": "AC", "ca": "AC", "na": "AN"}


def rules():
    """[(section, 'main'|'aux', first derived column, [expression nodes in column order])]"""
    cs, _ = build.all_constraints()
    _low, subs, num_main, num_aux = build.lower(cs)
    out, main_at, aux_at = [], num_main - sum(len(subs[s][0]) for s in SECTIONS), num_aux - sum(len(subs[s][1]) for s in SECTIONS)
    for sec in SECTIONS:
        for table, lst in (("main", subs[sec][0]), ("aux", subs[sec][1])):
            start = main_at if table == "main" else aux_at
            exprs = []
            for k, m in enumerate(lst):
                n = m.node   # x + (-1) * expr
                assert n.kind == "op" and n.op == "+" and n.lhs.kind == "in" and n.lhs.val[1] == start + k
                expr = Scheduler.is_neg(n.rhs)
                assert expr is not None
                exprs.append(expr)
            if exprs:
                out.append((sec, table, start, exprs))
            if table == "main":
                main_at += len(lst)
            else:
                aux_at += len(lst)
    assert (main_at, aux_at) == (num_main, num_aux)
    return out, num_main, num_aux


class Emitter:
    """Straight-line code for the rules of one section: common subexpressions once, `a + (-1)*b` as a subtraction."""

    def __init__(self, start, xfe_table):
        self.start, self.xfe_table = start, xfe_table
        self.lines, self.memo, self.is_base, self.n = [], {}, {}, 0

    def base(self, n):
        evaluates_to_base_element(n, self.is_base)
        return self.is_base[n.id]

    def leaf(self, n, dual):
        if n.kind == "b":
            return lit(mont(n.val))
        if n.kind == "x":
            return "xfe_make(" + ", ".join(lit(mont(c)) for c in n.val) + ")"
        if n.kind == "ch":
            return f"FCH({n.val})"
        kind, col = KIND[n.val[0]], n.val[1]
        own = (kind == "AC") == self.xfe_table and kind in ("MC", "AC")
        if own and col >= self.start:      # derived earlier in this section, same row: still in a register
            return f"d{col}"
        assert kind in ("MC", "AC") or dual, "successor rows only in the transition section"
        assert not (kind in ("MN", "AN") and own and col >= self.start)
        return f"F{kind}({col})"

    def emit(self, n, dual):
        if n.kind != "op":
            return self.leaf(n, dual)
        if n.id in self.memo:
            return self.memo[n.id]
        a, b, is_sub = n.lhs, n.rhs, False
        if n.op == "+":
            neg_r, neg_l = Scheduler.is_neg(n.rhs), Scheduler.is_neg(n.lhs)
            if neg_r is not None:
                a, b, is_sub = n.lhs, neg_r, True
            elif neg_l is not None:
                a, b, is_sub = n.rhs, neg_l, True
        ea, eb = self.emit(a, dual), self.emit(b, dual)
        ab, bb = self.base(a), self.base(b)
        if n.op == "+":
            if ab and bb:
                fn = "bfe_sub" if is_sub else "bfe_add"
            elif not ab and not bb:
                fn = "xfe_sub" if is_sub else "xfe_add"
            elif not ab and bb:
                fn = "xfe_sub_bfe" if is_sub else "xfe_add_bfe"
            elif is_sub:
                fn = "xfe_bfe_minus"
            else:
                fn, ea, eb = "xfe_add_bfe", eb, ea
        else:
            if ab and bb:
                fn = "bfe_mul"
            elif not ab and not bb:
                fn = "xfe_mul"
            elif not ab and bb:
                fn = "xfe_mul_bfe"
            else:
                fn, ea, eb = "xfe_mul_bfe", eb, ea
        name = f"t{self.n}"
        self.n += 1
        ty = "u64" if self.base(n) else "xfe"
        self.lines.append(f"    const {ty} {name} = {fn}({ea}, {eb});")
        self.memo[n.id] = name
        return name


def render(rule_list, num_main, num_aux):
    lines = ["// GENERATED by tools/air/export_fill.py -- the degree-lowering fill (values of the derived columns). Do not edit.",
             "// Rules: the substitutions of the deterministic degree lowering (tools/air), evaluated as the reference's",
             "// generated DegreeLoweringTable does (triton-constraint-builder/src/substitutions.rs:128-400).",
             '#include "fill.h"', "", "namespace tvm {"]
    kernels = []
    for sec, table, start, exprs in rule_list:
        dual = sec == "tran"
        xfe_table = table == "aux"
        em = Emitter(start, xfe_table)
        name = f"k_fill_{table}_{sec}"
        body = []
        for k, e in enumerate(exprs):
            v = em.emit(e, dual)
            body += em.lines
            em.lines = []
            col = start + k
            ty = "xfe" if xfe_table else "u64"
            if xfe_table:
                assert not em.base(e), "an auxiliary derived column holds an extension-field value"
            else:
                assert em.base(e)
            body.append(f"    const {ty} d{col} = {v};")
            body.append(f"    FILL_STORE_{'X' if xfe_table else 'B'}({col}, d{col});")
        lines += [f"// section {sec}: {table} columns {start} .. {start + len(exprs) - 1}",
                  f"__global__ void __launch_bounds__(256) {name}(FillArgs a) {{",
                  f"    FILL_PROLOGUE({1 if dual else 0}, {start}, {len(exprs)}, {'3' if xfe_table else '1'});"]
        lines += body
        lines += ["}", ""]
        kernels.append((name, table, sec))
    lines += ["}  // namespace tvm", ""]
    hdr = ["// GENERATED by tools/air/export_fill.py. Do not edit.", "#pragma once", '#include "fill.h"', "",
           f"#define TVM_FILL_NUM_MAIN {num_main}", f"#define TVM_FILL_NUM_AUX {num_aux}", "", "namespace tvm {"]
    for name, table, sec in kernels:
        hdr.append(f"__global__ void {name}(FillArgs a);")
    for table in ("main", "aux"):
        ks = [name for name, t, _ in kernels if t == table]
        hdr.append(f"typedef void (*FillKernel)(FillArgs);" if table == "main" else "")
        hdr.append(f"static FillKernel const TVM_FILL_{table.upper()}_KERNELS[{max(len(ks), 1)}] = {{" + ", ".join(ks or ["nullptr"]) + "};")
        hdr.append(f"#define TVM_FILL_{table.upper()}_NUM_KERNELS {len(ks)}")
    hdr += ["}  // namespace tvm", ""]
    return "\n".join(lines), "\n".join(hdr)


def oracle_rules(rule_list):
    """plain-data form for the CPU oracle: every expression as a nested tuple"""
    def conv(n):
        if n.kind == "b":
            return ("b", n.val)
        if n.kind == "x":
            return ("x", tuple(n.val))
        if n.kind == "ch":
            return ("ch", n.val)
        if n.kind == "in":
            return (KIND[n.val[0]], n.val[1])
        return (n.op, conv(n.lhs), conv(n.rhs))
    out = ["# GENERATED by tools/air/export_fill.py -- the substitution rules of the degree lowering as plain data.",
           "# (section, table, first derived column, [expression, ...]); expression = ('+'|'*', lhs, rhs) |",
           "# ('b', canonical value) | ('x', (c0, c1, c2)) | ('ch', k) | ('MC'|'MN'|'AC'|'AN', column)", "RULES = ["]
    for sec, table, start, exprs in rule_list:
        out.append(f"    ({sec!r}, {table!r}, {start}, [")
        for e in exprs:
            out.append(f"        {conv(e)!r},")
        out.append("    ]),")
    out.append("]")
    return "\n".join(out) + "\n"


def main():
    rule_list, num_main, num_aux = rules()
    src, hdr = render(rule_list, num_main, num_aux)
    csrc = os.path.join(ROOT, "triton_vm_amd", "csrc")
    open(os.path.join(csrc, "fill_gen.hip"), "w").write(src)
    open(os.path.join(csrc, "fill_gen.h"), "w").write(hdr)
    open(os.path.join(ROOT, "oracle", "degree_lowering_rules.py"), "w").write(oracle_rules(rule_list))
    for sec, table, start, exprs in rule_list:
        print(sec, table, start, len(exprs))


if __name__ == "__main__":
    main()
