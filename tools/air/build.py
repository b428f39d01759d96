"""Rebuild the reference's lowered AIR multicircuit and export it.

Mirrors /root/reference/triton-vm/build.rs:12-25 and
/root/reference/triton-constraint-builder/src/lib.rs:39-184:
  Constraints::all()  ->  lower_to_target_degree_through_substitutions  ->
  combine_with_substitution_induced_constraints  ->  (per section) base-field constraints first, then
  extension-field ones (codegen.rs:210-252).
"""
from . import cascade, cross_table, hash as hash_table, jump_stack, lookup, op_stack, processor, program, ram, u32
from .circuit import Builder, evaluates_to_base_element, lower_to_degree, num_visible_nodes, multicircuit_degree
from .defs import NUM_AUX_COLUMNS, NUM_MAIN_COLUMNS, TARGET_DEGREE

TABLES = [("Program", program), ("Processor", processor), ("OpStack", op_stack), ("Ram", ram),
          ("JumpStack", jump_stack), ("Hash", hash_table), ("Cascade", cascade), ("Lookup", lookup), ("U32", u32),
          ("GrandCrossTableArg", cross_table)]
SECTIONS = [("init", "initial_constraints", False), ("cons", "consistency_constraints", False),
            ("tran", "transition_constraints", True), ("term", "terminal_constraints", False)]


def all_constraints(tables=TABLES):
    """Constraints::all (lib.rs:39-129): one builder per section, tables in canonical order."""
    out, per_table = {}, {}
    for sec, fn, dual in SECTIONS:
        b = Builder(dual=dual)
        roots, counts = [], []
        for name, mod in tables:
            cs = getattr(mod, fn)(b)
            counts.append((name, len(cs)))
            roots += cs
        out[sec] = (b, roots)
        per_table[sec] = counts
    return out, per_table


def lower(constraints, target_degree=TARGET_DEGREE, num_main=NUM_MAIN_COLUMNS, num_aux=NUM_AUX_COLUMNS):
    """lower_to_target_degree_through_substitutions (lib.rs:131-171) + combine (lib.rs:174-184)."""
    result, subs = {}, {}
    for sec, _, _ in SECTIONS:
        b, roots = constraints[sec]
        main_subs, aux_subs = lower_to_degree(roots, b, target_degree, num_main, num_aux)
        num_main += len(main_subs)
        num_aux += len(aux_subs)
        subs[sec] = (main_subs, aux_subs)
        result[sec] = (b, roots + main_subs + aux_subs)
    return result, subs, num_main, num_aux


def ordered_roots(roots):
    """codegen.rs:210-252: constraints that evaluate to a base-field element first, then the rest."""
    nodes = [r.node for r in roots]
    base = [n for n in nodes if evaluates_to_base_element(n)]
    ext = [n for n in nodes if not evaluates_to_base_element(n)]
    return base, ext


def stats(constraints):
    return {sec: (len(roots), num_visible_nodes(roots), multicircuit_degree([r.node for r in roots]))
            for sec, (_, roots) in constraints.items()}
