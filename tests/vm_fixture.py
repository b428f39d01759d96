"""Valid master tables for the AIR / extend tests, produced by the oracle-side VM (oracle/vm): the tiny program of
`current_proof_version_is_still_current` (/root/reference/triton-vm/src/proof.rs:200-226) and
`program_executing_every_instruction` (/root/reference/triton-vm/src/stark.rs:4639-4803)."""
import functools
import os

import numpy as np

from oracle import degree_lowering as dlo
from oracle import oracle as orc
from oracle.vm import isa, tables as T, vm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY_PROGRAM = "pick 11 pick 12 pick 13 pick 14 pick 15 read_io 5 assert_vector halt"
# more of the programs the reference proves and verifies (stark.rs:4257-4317, triton-dev-util/src/example_programs.rs:70-96)
MANY_U32_PROGRAM = """
    push 1311768464867721216 split
    push 13387 push 78810 lt
    push 5 push 7 pow
    push 69584 push 6796 xor
    push 64972 push 3915 and
    push 98668 push 15787 div_mod
    push 15787 push 98668 div_mod
    push 98141 push 7397 and
    push 67749 push 60797 lt
    push 49528 split
    push 53483 call lsb
    push 79655 call is_u32
    push 60615 log_2_floor
    push 13 push 5 pow
    push 86323 push 37607 xor
    push 32374 push 20636 pow
    push 97416 log_2_floor
    push 14392 push 31589 div_mod
    halt
    lsb: push 2 swap 1 div_mod return
    is_u32: split pop 1 push 0 eq return
"""
PICK_AND_PLACE_PROGRAM = """
    read_io 5 read_io 5 read_io 4
    pick 2 pick 9 place 13 place 13
    pick 0 pick 7 place 13 place 13
    pick 2 pick 8 place 13 place 13
    pick 3 pick 4 place 13 place 13
    pick 0 pick 3 place 13 place 13
    pick 0 pick 3 place 13 place 13
    pick 1 pick 1 place 13 place 13
    write_io 5 write_io 5 write_io 4
    halt
"""
PICK_AND_PLACE_INPUT = [6, 3, 7, 5, 1, 2, 4, 4, 7, 3, 6, 1, 5, 2]
from oracle.vm.workload import FIBONACCI_PROGRAM, RAM_LOOP_PROGRAM, SPONGE_LOOP_PROGRAM, U32_LOOP_PROGRAM  # noqa: E402,F401
PROGRAMS = {"halt": ("halt", []), "many_u32": (MANY_U32_PROGRAM, []), "pick_and_place": (PICK_AND_PLACE_PROGRAM, PICK_AND_PLACE_INPUT)}


def hash_pair(left, right):
    return [int(v) for v in orc.from_mont(orc.hash_pair(orc.to_mont(left), orc.to_mont(right)))]


def non_determinism(which):
    """-> (secret input, secret digests, initial RAM) of the program's NonDeterminism"""
    if which != "every":
        return [], [], None
    node_5, node_4, node_3 = [5] * 5, [4] * 5, [3] * 5                                  # stark.rs:4770-4786
    node_2 = hash_pair(node_4, node_5)
    node_1 = hash_pair(node_2, node_3)
    ram = {i: 42 + i for i in range(1000)}
    ram.update({100_000 + i: v for i, v in enumerate(node_3)})
    return list(reversed(node_1)) + [1337] * 10, [node_4], ram


def run(which):
    """-> (program, aet, public input, public output); which: "tiny", "every", ("fib", index), ("u32", iterations), ("ram", iterations)
    or a key of PROGRAMS"""
    if which in PROGRAMS:
        text, public_input = PROGRAMS[which]
        program = isa.parse(text)
        aet, output = vm.trace_execution(program, public_input)
        return program, aet, list(public_input), output
    if isinstance(which, tuple) and which[0] in ("fib", "u32", "ram", "sponge"):
        program = isa.parse({"fib": FIBONACCI_PROGRAM, "u32": U32_LOOP_PROGRAM, "ram": RAM_LOOP_PROGRAM,
                             "sponge": SPONGE_LOOP_PROGRAM}[which[0]])
        aet, output = vm.trace_execution(program, [which[1]])
        return program, aet, [which[1]], output
    if which == "tiny":
        program = isa.parse(TINY_PROGRAM)
        public_input = vm.hash_varlen(program.to_bwords())
        aet, output = vm.trace_execution(program, public_input)
        return program, aet, public_input, output
    with open(os.path.join(GOLDEN, "program_every_instruction.tasm")) as f:
        program = isa.parse(f.read())
    public_input = [5] * 5                                                              # node_5, stark.rs:4770
    aet, output = vm.trace_execution(program, public_input, *non_determinism(which))
    return program, aet, public_input, output


@functools.lru_cache(maxsize=None)
def valid_tables(which, seed=1):
    """The padded, extended, degree-lowered master tables of a real execution, as the prover's hot path receives them:
    main [379][n], aux [91][n][3] (Montgomery words, column-major), challenges [63][3]."""
    program, aet, public_input, output = run(which)
    mt = T.MasterMainTable(aet).pad()
    n = mt.padded_height
    main = np.zeros((379, n), np.uint64)
    main[:T.NUM_MAIN] = orc.to_mont(np.array(mt.columns(), dtype=object))
    rng = np.random.default_rng(seed)
    sampled = [[int(v) for v in rng.integers(0, T.P, 3, dtype=np.uint64)] for _ in range(59)]
    challenges = T.derive_challenges(sampled, vm.hash_varlen(program.to_bwords()), public_input, output)
    aux = np.zeros((91, n, 3), np.uint64)
    aux[:T.NUM_AUX] = orc.to_mont(np.array(T.extend(mt.tables, challenges), dtype=object))
    aux[90] = orc.random_elements(rng, (n, 3))                                         # the batch-randomizer column
    ch = orc.to_mont(np.array(challenges, dtype=object))
    main, aux = dlo.fill(main, aux, ch)
    return main, aux, ch, mt


def constraint_violations(main, aux, challenges, rows=None):
    """(row, section) pairs on which some constraint of the oracle's (reference-pinned) AIR does not vanish:
    initial constraints on row 0, consistency on every row, transition on rows i, i+1, terminal on the last row
    (master_table.rs:1302-1359 divides exactly these out)."""
    n = main.shape[1]
    mr, ar = np.ascontiguousarray(main.T), np.ascontiguousarray(aux.transpose(1, 0, 2))
    bad = []
    for i in (range(n) if rows is None else rows):
        v = orc.air_constraint_values(mr[i], mr[(i + 1) % n], ar[i], ar[(i + 1) % n], challenges)
        sections = [("cons", 81, 178)]
        if i == 0:
            sections.append(("init", 0, 81))
        if i < n - 1:
            sections.append(("tran", 178, 581))
        if i == n - 1:
            sections.append(("term", 581, 604))
        bad += [(i, name) for name, a, b in sections if v[a:b].any()]
    return bad
