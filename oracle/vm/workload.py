"""TEST / WORKLOAD INFRASTRUCTURE (not part of the product): the programs of BASELINE.json's configurations and their
execution in the oracle-side VM (oracle/vm/vm.py, the stand-in for the reference's Rust VM -- trace generation is host work
in the reference too and lies outside `Prover::prove`).  `execution(kind, log2_padded_height)` returns what
`Prover::prove(claim, aet)` takes: the algebraic execution trace in the C ABI's layout and the claim's parts.  bench.py uses it
to build its input OUTSIDE the timed region; the tests use it for the full-size prove-and-verify cases."""
import numpy as np

# the program of the reference's headline benchmark, prove_fib (benches/prove_fib.rs:8-24 runs it with index 100;
# triton-dev-util/src/example_programs.rs:6-38): ten instructions per iteration
FIBONACCI_PROGRAM = """
    push 0 push 1 read_io 1
    dup 0 skiz call fib_loop
    pop 1 write_io 1 halt
    fib_loop:
        push -1 add swap 2 dup 1 add swap 1 swap 2 dup 0 skiz recurse return
"""


# a loop of u32 operations on fresh operand pairs: every iteration adds a 33-row section to the U32 table (BASELINE.json's
# "many_u32_ops at 2^20 rows": 31775 iterations fill 1 048 575 rows)
U32_LOOP_PROGRAM = """
    read_io 1
    call loop
    pop 1 halt
    loop:
        dup 0 push 2147483648 add
        dup 1 xor pop 1
        push -1 add dup 0 skiz recurse return
"""
# a loop that writes to a fresh RAM address every iteration: as many distinct RAM pointers as iterations (the RAM table's
# Bezout coefficient polynomials have that many coefficients)
RAM_LOOP_PROGRAM = """
    read_io 1
    call loop
    pop 1 halt
    loop:
        dup 0 dup 0 mul
        dup 1 push 1000 mul
        write_mem 1 pop 1
        push -1 add dup 0 skiz recurse return
"""
# a loop of sponge operations: every iteration squeezes and absorbs (two Tip5 permutations = 12 rows of the hash table
# against 8 processor cycles), so the hash table sets the padded height -- the hash-heavy shape that stands in for
# BASELINE.json's recursive-verifier program (which lives outside the reference repository)
SPONGE_LOOP_PROGRAM = """
    read_io 1
    sponge_init
    call loop
    pop 1 halt
    loop:
        sponge_squeeze sponge_absorb
        push -1 add dup 0 skiz recurse return
"""
LOOP_PROGRAMS = {"fib": FIBONACCI_PROGRAM, "u32": U32_LOOP_PROGRAM, "ram": RAM_LOOP_PROGRAM, "sponge": SPONGE_LOOP_PROGRAM}


def loop_index(kind, log2_padded_height):
    """the public input (iteration count) that makes the padded height exactly 2^log2: fib runs ten processor cycles per
    iteration, u32 adds 33 rows of the U32 table, ram 14 cycles, sponge 24 rows of the hash table per iteration"""
    n = 1 << log2_padded_height
    return {"u32": n // 33, "ram": (n - 20) // 14, "sponge": (n - 40) // 24, "fib": (n - 20) // 10}[kind]


def execution(kind, log2_padded_height, host_bezout=True):
    """run `kind` (fib | u32 | ram | sponge) so that the padded height is 2^log2_padded_height ->
    dict(aet=arrays for tvm_aet, padded_height, program_digest, public_input, public_output (Montgomery words), cycles,
    table_heights, program)"""
    from oracle import oracle as orc
    from oracle.vm import isa, vm
    from oracle.vm.aet_export import aet_arrays

    index = loop_index(kind, log2_padded_height)
    program = isa.parse(LOOP_PROGRAMS[kind])
    aet, output = vm.trace_execution(program, [index])
    padded_height = aet.padded_height()
    if padded_height != 1 << log2_padded_height:
        raise ValueError(f"{kind} with input {index} pads to {padded_height} rows, not 2^{log2_padded_height}")
    mont = lambda values: orc.to_mont(np.array(values, dtype=object)) if len(values) else np.zeros(0, np.uint64)
    return dict(aet=aet_arrays(orc, aet, host_bezout), padded_height=padded_height,
                program_digest=orc.hash_varlen(mont(program.to_bwords())), public_input=mont([index]), public_output=mont(output),
                cycles=aet.height_of_table("Processor"), index=index, program=program,
                table_heights={name: aet.height_of_table(name) for name in ("Processor", "OpStack", "Ram", "U32", "Hash", "Cascade")})
