"""TEST / WORKLOAD INFRASTRUCTURE (not part of the product): the oracle-side VM's algebraic execution trace in the layout the
C ABI's `tvm_aet` takes, i.e. as the reference's AlgebraicExecutionTrace holds it (/root/reference/triton-vm/src/aet.rs:41-96):
Montgomery words, row-major.  Used by the tests and by bench.py's workload generation (the VM run that stands in for the
reference's Rust VM happens outside every timed region)."""
import numpy as np


def aet_arrays(orc, aet, host_bezout=True):
    """the oracle VM's AET in the layout of AlgebraicExecutionTrace (aet.rs:41-96): Montgomery words, row-major.
    host_bezout=False leaves the RAM table's Bezout coefficient polynomials to the device."""
    from oracle.vm import tables as T

    def M(rows, width):
        if not rows:
            return np.zeros((0, width), np.uint64)
        try:
            canonical = np.array(rows, dtype=np.uint64)       # the VM keeps canonical values: one pass, also at 2^20 rows
        except OverflowError:
            canonical = np.array([[v % T.P for v in r] for r in rows], dtype=object)
        return orc.to_mont(canonical.reshape(-1, width))

    hash_rows = lambda trace: [T.hash_table_row(0, ci, rnd, state) for ci, rnd, state in trace]   # Mode is set by fill
    b0 = b1 = []
    if host_bezout:
        ram_rows, _ = (T.fill_ram(aet) if aet.ram_trace else ([], []))
        unique = list(dict.fromkeys(r[T.M["Ram"]["RamPointer"]] for r in ram_rows))
        b0, b1 = T.bezout_coefficient_polynomials_coefficients(unique)
    arrays = {
        "program_words": orc.to_mont(np.array(aet.program.to_bwords(), dtype=object)),
        "instruction_multiplicities": np.array(aet.instruction_multiplicities, np.uint32),
        "processor_trace": M(aet.processor_trace, 39),
        "op_stack_trace": M(aet.op_stack_underflow_trace, 4),
        "ram_trace": M([list(r) + [0, 0, 0] for r in aet.ram_trace], 7),
        "bezout_coefficients_0": orc.to_mont(np.array(b0, dtype=object)) if b0 else np.zeros(0, np.uint64),
        "bezout_coefficients_1": orc.to_mont(np.array(b1, dtype=object)) if b1 else np.zeros(0, np.uint64),
        "program_hash_trace": M(hash_rows(aet.program_hash_trace), 67),
        "sponge_trace": M(hash_rows(aet.sponge_trace), 67),
        "hash_trace": M(hash_rows(aet.hash_trace), 67),
        "u32_entries": np.array([[T.OP[name], int(orc.to_mont([lhs])[0]), int(orc.to_mont([rhs])[0]), mult]
                                 for (name, lhs, rhs), mult in aet.u32_entries.items()], np.uint64).reshape(-1, 4),
        "cascade_entries": np.array([[limb, mult] for limb, mult in aet.cascade_multiplicities.items()], np.uint64).reshape(-1, 2),
        "lookup_multiplicities": np.array(aet.lookup_multiplicities, np.uint64),
    }
    if not host_bezout:
        del arrays["bezout_coefficients_0"], arrays["bezout_coefficients_1"]
    return arrays
